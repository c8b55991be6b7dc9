#!/usr/bin/env python3
"""bench.py -- rollup-main tx-witnesses/sec on MI355X (BASELINE.json metric).

A "step" is one complete witness pass of RollupMain(nTx, nLevels, maxL1Tx, maxFeeTx) over
`--batches-per-launch` independent synthetic batches (one context, one set of kernel launches; the
value counts every transaction of every batch) whose inputs are already resident in HBM: DecodeTx + RollupTx for every
transaction, the fee transactions and HashInputs (SHA-256), every constraint checked. Steps are
issued round-robin over `--inflight` contexts/streams (independent batches in flight);
the timed region is bracketed by barrier + device synchronisation and includes the constraint
check of every step.

N > 1 (launched by torch.distributed.run, one process per GPU): every rank runs its own batches
-- batch-level data parallelism, no data-path collective (DESIGN.md "Multi-GPU") -- and the value
is all ranks' transactions over the max-over-ranks time ("scaling": "weak").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# independent batches are issued on separate HIP streams; give the runtime enough hardware queues
# for them to overlap (the ROCm default maps all streams onto 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_tx(L, F):
    # 32 B x (DecodeTx + RollupTx R1CS variables, reference tools/circuit-constraints.js:31-44) + packed inputs
    packed = {(32, 64): 4312, (16, 64): 3288}.get((L, F), 40 * 32 + 2 * (L + 1) * 32 + F * 24)
    return 32 * ((4 * L + 1473) + (974 * L + 14552 + 5 * F)) + packed


def poseidon_rates(L, torch):
    """BASELINE.json's second metric, Poseidon-BN254/sec: 2^20 permutations per launch, digest-only and with the
    S-box witness (the circuit's Poseidon signals), HIP events on the launch stream."""
    out = {}
    n = 1 << 20
    s = torch.cuda.current_stream().cuda_stream
    for t in (3, 5):
        nsbox = 8 * t + [56, 57, 56, 60, 60, 63][t - 2]
        g = torch.Generator(device="cpu").manual_seed(t)
        x = torch.randint(0, 2**31 - 1, (n * (t - 1), 8), dtype=torch.int32, generator=g)
        x[:, 7] &= 0x0FFFFFFF  # < 2^252 < r
        d_in = x.cuda()
        d_out = torch.empty((n, 8), dtype=torch.int32, device="cuda")
        d_wit = torch.empty((3 * nsbox * n, 8), dtype=torch.int32, device="cuda")
        for mode, wit in (("digest", None), ("witness", d_wit.data_ptr())):
            L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), wit, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                L.poseidon_batch_dev(t, n, d_in.data_ptr(), d_out.data_ptr(), wit, s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            by = n * (32 * (t - 1) + 32 + (96 * nsbox if wit else 0))
            out["t%d_%s" % (t, mode)] = {"perm_per_s": round(n / ms * 1e3, 0), "GBs": round(by / ms / 1e6, 1)}
        del d_in, d_out, d_wit
    return out


def measured_traffic(kernel, bpl):
    """HBM bytes of one launch of `kernel` (mean over its launches of the transaction grid) from the committed rocprofv3 PMC passes
    (profiles/r01_hbm_counters.json, collected by tools/profile.sh on the same command line); None when that file does not cover
    this configuration."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_counters.json")))
        if d.get("batches_per_launch") != bpl:
            return None
        k = d["kernels"][kernel]
        if "fetch_bytes_mean" in k:   # per launch like `achieved`: mean over the kernel's launches of the transaction grid
            return int(k["fetch_bytes_mean"] + k["write_bytes_mean"])
        return int(k["fetch_bytes"] + k["write_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(n_tx, L, max_l1, F, workers):
    """The CPU oracle (restated CPU path, kind "port") on a bounded sample of the same workload: `workers` processes, one batch
    of `n_tx` transactions each, started together (tests/cpu_baseline_worker.py); value = all their transactions / the slowest run."""
    import subprocess
    script = os.path.join(ROOT, "tests", "cpu_baseline_worker.py")
    start = time.time() + 6.0 + 0.012 * n_tx   # every worker has built its batch by then
    procs = [subprocess.Popen([sys.executable, script, str(n_tx), str(L), str(max_l1), str(F), repr(start)], stdout=subprocess.PIPE, text=True)
             for _ in range(workers)]
    outs = [p.communicate()[0].split() for p in procs]
    if any(p.returncode != 0 for p in procs):
        raise RuntimeError("cpu_baseline worker failed")
    times = [float(o[0]) for o in outs]
    late = max(float(o[1]) for o in outs)
    dt = max(times) + late   # a late starter only makes the denominator larger
    return {"value": round(workers * n_tx / dt, 2), "unit": "tx-witnesses/s", "cores": workers, "kind": "port",
            "per_core": round(n_tx / (sum(times) / len(times)), 2),
            "sample": "RollupMain(nTx=%d,nLevels=%d,maxL1Tx=%d,maxFeeTx=%d): %d processes x one batch, %.1f s, host has %d logical CPUs"
                      % (n_tx, L, max_l1, F, workers, dt, os.cpu_count() or 0)}


def bench_sharded(args, L, bb, inp, rank, world, local, n_l2):
    """One batch sharded by transaction index over `world` GPUs (circuits_amd/multigpu.py)."""
    import torch
    import torch.distributed as dist
    from circuits_amd.multigpu import ShardedBatch
    nTx, lv, m1, F = args.nTx, args.nLevels, args.maxL1Tx, args.maxFeeTx
    c = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=local)
    c.set_inputs(inp)
    stream = torch.cuda.Stream(device=local)

    def alloc(n):
        return torch.zeros(n, dtype=torch.uint8, device="cuda")

    def all_gather(recv, send):
        if world == 1:
            recv.copy_(send)
            return
        with torch.cuda.stream(stream):
            dist.all_gather_into_tensor(recv, send)

    sb = ShardedBatch(c, L, nTx, rank, world, alloc, all_gather)
    sb.step(stream.cuda_stream)
    if rank == 0 and not args.no_verify:
        assert c.get("main.hashGlobalInputs") == bb.get_hash_inputs(), "hashGlobalInputs mismatch"
    for _ in range(args.warmup):
        sb.step(stream.cuda_stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sb.step(stream.cuda_stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        value = nTx * args.steps / dt
        abytes = algorithmic_bytes_per_tx(lv, F) * nTx
        print(json.dumps({
            "metric": "rollup-main tx-witnesses/sec (nTx=%d, nLevels=%d)" % (nTx, lv), "value": round(value, 1), "unit": "tx-witnesses/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32x9 (254-bit Montgomery Fr, 29-bit limbs, integer)",
            "data": "synthetic",
            "config": {"workload": "rollup-main nTx=%d nLevels=%d maxL1Tx=%d maxFeeTx=%d" % (nTx, lv, m1, F), "parallelism": "tx-shard%d" % world,
                       "collective": "one all_gather of %d B per step" % (sb.slot * world), "l1_txs": nTx - n_l2, "l2_signed_txs": n_l2},
            "roofline": {"bound": "hbm", "kernel": "whole sharded pass", "achieved": round(abytes / (dt / args.steps) / 1e9, 2), "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": round(abytes / (dt / args.steps) / 1e9 / (HBM_PEAK_GBS * world), 5), "traffic": None}}))
    if world > 1:
        dist.destroy_process_group()


def bench_withdraw(args, L, rank, world, local):
    """BASELINE config 5: 2^20 independent Withdraw(nLevels) witnesses (SMTVerifier + HashState + 2-block SHA-256 bit witness),
    `--withdraw-per-launch` instances per kernel launch, exits drawn from one exit tree. One lane per witness."""
    import torch
    import torch.distributed as dist
    from circuits_amd import builder as B
    lv = args.nLevels
    n_leaves = 256
    db = B.RollupDB(chain_id=1)
    keys = [B.Account(900 + i) for i in range(8)]
    nTx = 2 * n_leaves
    bb = db.build_batch(nTx, lv, n_leaves, 1)
    for i in range(n_leaves):
        a = keys[i % 8]
        bb.add_tx({"fromIdx": 0, "loadAmountF": B.fix2float(1000 + i), "tokenID": 1, "fromBjjCompressed": a.bjj_compressed, "fromEthAddr": a.eth_addr, "toIdx": 0, "onChain": 1})
    bb.build()
    bb2 = db.build_batch(nTx, lv, n_leaves, 1)
    for i in range(n_leaves):
        bb2.add_tx({"fromIdx": 256 + i, "toIdx": 1, "amount": 10 + i, "tokenID": 1, "userFee": 0, "onChain": 0, "signer": keys[i % 8]})
    bb2.build()
    ins = [B.withdraw_input(bb2, 256 + i, lv) for i in range(n_leaves)]
    N = args.withdraw_per_launch
    free_b, _ = torch.cuda.mem_get_info()
    probe = L.ctx("withdraw", nLevels=lv, device=local, n_instances=1)
    wl = probe.witness_len()
    del probe
    N = max(64, min(N, int((free_b - (6 << 30)) // (wl * 32 * 1.02)) // 64 * 64))
    c = L.ctx("withdraw", nLevels=lv, device=local, n_instances=N)
    names = [n for n, _ in c.input_names()]
    for name in names:
        rows = [ins[k % n_leaves][0][name] for k in range(N)]
        c.set_input(name, rows, instance=-1)
    stream = torch.cuda.Stream(device=local)
    c.set_profiling(True)
    c.enqueue(stream.cuda_stream)
    c.check()
    for k in (0, N - 1):
        assert c.get("main.hashGlobalInputs", k) == ins[k % n_leaves][1], "hashGlobalInputs mismatch"
    c.set_profiling(False)
    launches = max(1, (args.withdraw_total + N - 1) // N)
    for _ in range(args.warmup):
        c.enqueue(stream.cuda_stream)
        c.check()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(launches):
            c.enqueue(stream.cuda_stream)
        c.check()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        total = launches * N * args.steps * world
        abytes = wl * 32 * N
        ms_kernel = dt / (args.steps * launches) * 1e3   # the two kernels of a launch run concurrently: wall time per launch
        print(json.dumps({
            "metric": "withdraw witnesses/sec (nLevels=%d)" % lv, "value": round(total / dt, 1), "unit": "witnesses/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32x9 (254-bit Montgomery Fr, 29-bit limbs, integer)", "data": "synthetic",
            "config": {"workload": "withdraw nLevels=%d, %d witnesses per step in %d launches of %d" % (lv, launches * N, launches, N),
                       "witness_elements": wl, "distinct_exit_leaves": n_leaves},
            "roofline": {"bound": "hbm", "kernel": "k_withdraw_sha || k_withdraw", "achieved": round(abytes / (ms_kernel * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(abytes / (ms_kernel * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "launch_ms": round(ms_kernel, 3),
                         "algorithmic_bytes_per_launch": int(abytes)}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nTx", type=int, default=2048)
    ap.add_argument("--nLevels", type=int, default=32)
    ap.add_argument("--maxL1Tx", type=int, default=256)
    ap.add_argument("--maxFeeTx", type=int, default=64)
    ap.add_argument("--accounts", type=int, default=0, help="accounts in the synthetic state before the batch (default 4 * nTx, the reference recipe)")
    ap.add_argument("--inflight", type=int, default=2, help="contexts in flight (each with its own witness buffers and streams): the fee/SHA tail of one step overlaps the next step's kernels")
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU-baseline processes (0 = min(16, logical CPUs); 64 processes were measured slower in total: 867 vs 956 tx/s)")
    ap.add_argument("--cpu-sample", type=int, default=768, help="nTx of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--batches-per-launch", type=int, default=32,
                    help="independent batches evaluated by ONE set of kernel launches (context with n_instances = B): more wavefronts per launch")
    ap.add_argument("--workload", choices=["rollup-main", "withdraw"], default="rollup-main",
                    help="withdraw = BASELINE config 5 (2^20 independent Withdraw(nLevels) witnesses)")
    ap.add_argument("--withdraw-total", type=int, default=1 << 20)
    ap.add_argument("--withdraw-per-launch", type=int, default=1 << 16)
    ap.add_argument("--latency-scheduling", action="store_true",
                    help="with --inflight 1: HZ_FLAG_LATENCY contexts (concurrent kernel chains on disjoint compute units); for the "
                         "single-batch latency figure: --batches-per-launch 1 --inflight 1 --latency-scheduling")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-poseidon", action="store_true", help="skip the Poseidon-BN254/sec secondary metric")
    ap.add_argument("--calibrate-copy", action="store_true",
                    help="profiling aid: one 1 GiB device-to-device tensor copy before the timed region, a known byte count that "
                         "calibrates the FETCH_SIZE / WRITE_SIZE counters of a rocprofv3 --pmc pass (tools/profile.sh)")
    ap.add_argument("--shard-tx", action="store_true",
                    help="BASELINE config 4: shard ONE batch by transaction index over the ranks (one RCCL all_gather of the "
                         "data-availability records, FeeTx + HashInputs on rank 0); strong scaling. Default: independent batches per rank.")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "HZ_BENCH_DEVICE" in os.environ:   # test hook: several ranks on one GPU (with HZ_BENCH_BACKEND=gloo)
        local = int(os.environ["HZ_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(os.environ.get("HZ_BENCH_BACKEND", "nccl"))
    torch.cuda.set_device(local)

    from circuits_amd import lib
    from circuits_amd import builder as B
    L = lib()
    if L.device_count() <= 0:
        raise SystemExit("no gfx950 device")
    if args.workload == "withdraw":
        return bench_withdraw(args, L, rank, world, local)
    nTx, lv, m1, F = args.nTx, args.nLevels, args.maxL1Tx, args.maxFeeTx
    # synthetic batch (reference tools/generate-input.js recipe, SURVEY 8d: a state of 4*nTx accounts), same seed on every rank;
    # built with the device batch builder (circuits_amd/builder.py DagHasher: the tree hashing of the 4*nTx + maxL1Tx accounts and of
    # the batch in ~60 launches). The state size matters: the Merkle proofs of a tree of 2^13 leaves reach their leaf at level 13-14,
    # the levels below hash empty subtrees.
    n_acc = args.accounts if args.accounts > 0 else 4 * nTx
    bb = B.synthetic_batch(nTx, lv, m1, F, n_accounts=n_acc, seed=0x48455A31, device=local)
    inp = bb.get_input()
    n_l2 = sum(1 for x in inp["onChain"] if not x)
    if args.shard_tx:
        return bench_sharded(args, L, bb, inp, rank, world, local, n_l2)
    Bp = max(1, args.batches_per_launch)
    # more than 4 contexts (3 streams each) exhaust the runtime's per-queue scratch reservations (HSA_STATUS_ERROR_OUT_OF_RESOURCES)
    inflight = max(1, min(args.inflight, 4, args.steps if args.steps > 0 else 1))
    # every batch keeps its whole witness resident (3.86 GB at the default shape): fit batches x contexts into free HBM
    probe = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=local, n_instances=1)
    per_batch = probe.witness_len() * 32 * 1.04 + (64 << 20)
    del probe
    free_b, _total_b = torch.cuda.mem_get_info()
    fit = int((free_b - (6 << 30)) // (per_batch * inflight))
    if fit < Bp:
        print("bench: %d batches x %d contexts do not fit %.0f GB of free HBM, using %d batches per launch" % (Bp, inflight, free_b / 1e9, max(1, fit)), file=sys.stderr)
        Bp = max(1, fit)
    ctxs, streams = [], []
    for k in range(inflight):
        c = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=local, n_instances=Bp,
                  flags=2 if (args.latency_scheduling and inflight == 1) else 0)
        c.set_inputs(inp, instance=0)  # inputs resident in HBM before the timed region
        for b in range(1, Bp):
            c.copy_instance_inputs(0, b)  # same synthetic batch in every instance (device-to-device)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(device=local))
    # one checked pass (parity with the builder's independently computed public output)
    ctxs[0].set_profiling(True)
    ctxs[0].enqueue(streams[0].cuda_stream)
    ctxs[0].check()
    if not args.no_verify:
        for b in (0, Bp - 1):
            assert ctxs[0].get("main.hashGlobalInputs", b) == bb.get_hash_inputs(), "hashGlobalInputs mismatch"
    ctxs[0].set_profiling(False)

    def run_steps(n):
        pending = [False] * inflight
        for i in range(n):
            k = i % inflight
            if pending[k]:
                ctxs[k].check()
            ctxs[k].enqueue(streams[k].cuda_stream)
            pending[k] = True
        for k in range(inflight):
            if pending[k]:
                ctxs[k].check()

    if args.calibrate_copy:
        a = torch.zeros(1 << 28, dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        del a, b
    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # latency of one step alone on the device (wall clock around enqueue + check; `batches_per_launch` batches)
    lat = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ctxs[0].enqueue(streams[0].cuda_stream)
        ctxs[0].check()
        lat.append((time.perf_counter() - t1) * 1e3)
    single_ms = min(lat)
    # per-kernel device times (HIP events on the stream each kernel is launched on), each kernel alone
    # on the device (in the timed region above the EdDSA and fee-tx chains overlap the hash/SMT chain)
    ctxs[0].set_profiling(True, exclusive=True)  # each kernel alone on the device: durations for the per-kernel roofline
    acc = {}
    reps = 3
    for _ in range(reps):
        ctxs[0].enqueue(streams[0].cuda_stream)
        ctxs[0].check()
        for name, ms, by, units in ctxs[0].profile():
            a = acc.setdefault(name, [0.0, by, units, 0])
            a[0] += ms / reps      # a kernel launched in several pieces (the SMT chain: chunks of levels) adds up over its launches
            a[3] += 1.0 / reps
    ctxs[0].set_profiling(False)

    if rank == 0:
        total_tx = nTx * Bp * args.steps * world
        value = total_tx / dt
        # dominant kernel = most GPU time per step over all its launches (k_smt runs for the transactions and for the
        # fee transactions); its roofline is quoted on the transaction launch
        kern = {"smt": "k_smt", "fee_smt": "k_smt", "hash4": "k_hash4", "fee_hash": "k_hash4", "eddsa": "k_eddsa", "eddsa_fix": "k_eddsa_fix", "front": "k_main_front"}
        tot = {}
        for name, v in acc.items():
            tot[kern.get(name, name)] = tot.get(kern.get(name, name), 0.0) + v[0]
        byt = {}
        for name, v in acc.items():
            byt[kern.get(name, name)] = byt.get(kern.get(name, name), 0) + v[1]
        # dominant kernel for an HBM roofline = the kernel that writes most of the witness (k_smt: three quarters of a step's
        # bytes, and the only one besides the front / hash kernels that fills the device). k_eddsa's single launch lasts longer
        # but runs on 384 wavefronts, latency bound, underneath the others: its figures are in kernels_ms / kernels_GBs.
        dk = max(tot, key=lambda k: byt[k])
        dname = max((n for n in acc if kern.get(n, n) == dk), key=lambda n: acc[n][0])
        dms, dbytes, dunits, dlaunches = acc[dname]
        dlaunches = max(1, int(round(dlaunches)))
        achieved = dbytes / (dms * 1e-3) / 1e9
        traffic = measured_traffic(dk, Bp)
        out = {
            "metric": "rollup-main tx-witnesses/sec (nTx=%d, nLevels=%d)" % (nTx, lv),
            "value": round(value, 1), "unit": "tx-witnesses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x9 (254-bit Montgomery Fr, 29-bit limbs, integer)", "data": "synthetic",
            "config": {"workload": "rollup-main nTx=%d nLevels=%d maxL1Tx=%d maxFeeTx=%d" % (nTx, lv, m1, F), "batches_per_launch": Bp, "contexts_in_flight": inflight,
                       "state_accounts": n_acc, "l1_txs": nTx - n_l2, "l2_signed_txs": n_l2, "parallelism": "batch-dp%d" % world,
                       "witness_bytes_per_batch": ctxs[0].witness_len() * 32, "step_latency_ms": round(single_ms, 3)},
            "roofline": {"bound": "hbm", "kernel": dk, "launch": dname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "launches_per_step": dlaunches, "launch_ms": round(dms / dlaunches, 3),
                         "algorithmic_bytes_per_launch": int(dbytes // dlaunches), "gpu_ms_per_step_all_launches": round(tot[dk], 3),
                         "note": "algorithmic bytes = 32 B x the witness signals this launch is responsible for (the SMT chain is launched in chunks "
                                 "of levels: per-launch figures are the mean over a step's launches); duration = HIP events on its stream with the "
                                 "kernel alone on the device; traffic = FETCH_SIZE + WRITE_SIZE of the committed PMC passes. "
                                 "Levels of an SMT proof below the leaf (the hash of an empty subtree) are stored from a constant block and are "
                                 "HBM-store bound; the levels that hash data are integer-VALU issue bound (DESIGN.md 4)"},
            "whole_pass": {"algorithmic_bytes_per_tx": algorithmic_bytes_per_tx(lv, F),
                           "achieved_GBs": round(algorithmic_bytes_per_tx(lv, F) * value / 1e9, 2)},
            "kernels_ms": {k: round(v[0], 3) for k, v in acc.items()},
            "kernels_GBs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in acc.items() if v[0] > 0},
        }
        if world == 1 and not args.no_poseidon:
            out["poseidon_bn254"] = poseidon_rates(L, torch)
        if world == 1 and args.cpu_sample > 0:
            workers = args.cpu_workers if args.cpu_workers > 0 else max(1, min(16, (os.cpu_count() or 1)))
            out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample, nTx), lv, min(m1, max(1, args.cpu_sample // 8)), F, workers)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
