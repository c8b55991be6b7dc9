#!/usr/bin/env python3
"""bench.py -- rollup-main tx-witnesses/sec on MI355X (BASELINE.json metric).

A "step" is one complete witness pass of RollupMain(nTx, nLevels, maxL1Tx, maxFeeTx) over
`--batches-per-launch` independent synthetic batches (one context, one set of kernel launches; the
value counts every transaction of every batch): DecodeTx + RollupTx for every transaction, the fee
transactions and HashInputs (SHA-256), every constraint checked. Steps are issued round-robin over
`--inflight` contexts/streams (independent batches in flight); the timed region is
bracketed by barrier + device synchronisation and includes the constraint check of every step.

Every instance of every context holds a DIFFERENT seeded batch (circuits_amd/batchgen.py builds them on the host cores before the
timed region). `value` is measured with the inputs resident in HBM; `value_e2e` repeats the run with every batch's packed inputs
uploaded from pinned host memory inside the timed region (hz_inputs_upload: the boundary the reference's calculateWitness(input)
really has, test/helpers/helpers.js:147-149).

`--gpus N` with N > 1 launches N ranks itself (torch.distributed.run, one process per GPU over RCCL) unless it already runs
under a launcher: every rank runs its own batches -- batch-level data parallelism, no data-path collective (DESIGN.md
"Multi-GPU") -- and the value is all ranks' transactions over the max-over-ranks time ("scaling": "weak"); the same line carries
BASELINE config 4, ONE batch sharded by transaction index with its single all_gather, as `shard_tx` (strong scaling).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# independent batches are issued on separate HIP streams; give the runtime enough hardware queues
# for them to overlap (the ROCm default maps all streams onto 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DTYPE = "u32x9 (254-bit Montgomery Fr, 29-bit limbs, integer)"
SEED = 0x48455A31


def algorithmic_bytes_per_tx(L, F):
    # 32 B x (DecodeTx + RollupTx R1CS variables, reference tools/circuit-constraints.js:31-44) + packed inputs
    packed = {(32, 64): 4312, (16, 64): 3288}.get((L, F), 40 * 32 + 2 * (L + 1) * 32 + F * 24)
    return 32 * ((4 * L + 1473) + (974 * L + 14552 + 5 * F)) + packed


class Dist:
    """rank / world / collectives for timing; backend "nccl" (= RCCL) unless the test hook asks for gloo"""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("HZ_BENCH_BACKEND", "nccl")
        if "HZ_BENCH_DEVICE" in os.environ:   # test hook: several ranks on one GPU (with HZ_BENCH_BACKEND=gloo)
            self.local = int(os.environ["HZ_BENCH_DEVICE"])
        import torch
        self.torch = torch
        torch.cuda.set_device(self.local)
        # HZ_BENCH_FORCE_COLLECTIVE=1 (test hook): a process group even at world size 1, so that the RCCL calls of the sharded
        # pass (all_gather_into_tensor / broadcast on the pass's stream) run on a box with one GPU
        self.force = self.world == 1 and os.environ.get("HZ_BENCH_FORCE_COLLECTIVE") == "1"
        if self.world > 1 or self.force:
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.force:
                s = socket.socket()
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
                s.close()
                dist.init_process_group(self.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
            else:
                dist.init_process_group(self.backend)
            self.dist = dist

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.backend == "gloo" else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn):
        """fn() bracketed by barrier + device synchronisation on both sides; max over ranks, seconds"""
        self.barrier()
        t0 = time.perf_counter()
        fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        if self.world > 1 or self.force:
            self.dist.destroy_process_group()


def poseidon_rates(L, torch):
    """BASELINE.json's second metric, Poseidon-BN254/sec: 2^20 permutations per launch, digest-only and with the
    S-box witness (the circuit's Poseidon signals), HIP events on the launch stream. Both forms are integer-issue bound
    (0.83-0.92 of the issue slots at 4.3 cycles per wave-instruction); the with-witness form also reports its store stream against
    the HBM peak (the contract's figure) and names the roofline with the larger fraction in `bound`."""
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
            r = {"perm_per_s": round(n / ms * 1e3, 0), "GBs": round(by / ms / 1e6, 1), "launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": by}
            # the integer-issue side: SQ_INSTS_VALU of this instantiation's launch of 2^20 permutations (committed PMC pass,
            # tools/gpu_suite.sh pmc-poseidon) x the issue cost of the mix / (SIMDs x clock x this run's launch time)
            insts = poseidon_valu("poseidon_batch_kernel<%d, %s>" % (t, "true" if wit else "false"))
            props = torch.cuda.get_device_properties(torch.cuda.current_device())
            fv = insts * VALU_CYCLES_PER_INST / (props.multi_processor_count * 4 * float(getattr(props, "clock_rate", 0) or 2400000) * 1e3 * ms * 1e-3) if insts else None
            if fv is not None:
                r["valu_insts_per_launch"], r["frac_valu"] = int(insts), round(fv, 5)
            if wit:
                fh = by / ms / 1e6 / HBM_PEAK_GBS
                r["roofline"] = {"bound": "valu" if fv is not None and fv > fh else "hbm", "kernel": "poseidon_batch_kernel<%d, witness>" % t, "achieved": r["GBs"],
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fh, 5), "frac_hbm": round(fh, 5), "frac_valu": r.get("frac_valu"),
                                 "traffic": measured_traffic("poseidon_t%d_witness" % t, None)[0], "traffic_source": measured_traffic("poseidon_t%d_witness" % t, None)[1]}
            out["t%d_%s" % (t, mode)] = r
        del d_in, d_out, d_wit
    return out


PROFILE_ROUNDS = ("r06", "r05", "r04")   # committed counter passes, newest first


def _profile_json(suffix):
    for r in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", "%s_%s" % (r, suffix))
        try:
            return json.load(open(path)), "profiles/%s_%s" % (r, suffix)
        except (OSError, ValueError):
            continue
    return None, None


def measured_traffic(kernel, bpl):
    """(HBM bytes of one launch of `kernel` (mean over its launches of the transaction grid), the file they come from): the committed
    rocprofv3 PMC passes (profiles/rNN_hbm_counters.json, collected by tools/profile.sh on the same command line) -- a recorded
    constant, not a measurement of this run; (None, None) when no file covers this configuration."""
    d, src = _profile_json("hbm_counters.json")
    try:
        if d is None or (bpl is not None and d.get("batches_per_launch") != bpl):
            return None, None
        k = d["kernels"][kernel]
        if "write_bytes_median" in k:   # the steady-state launch (the first ones into a fresh buffer store what the constant marks later leave)
            return int(k["fetch_bytes_median"] + k["write_bytes_median"]), src
        if "fetch_bytes_mean" in k:   # per launch like `achieved`: mean over the kernel's launches of the transaction grid
            return int(k["fetch_bytes_mean"] + k["write_bytes_mean"]), src
        return int(k["fetch_bytes"] + k["write_bytes"]), src
    except (KeyError, ValueError):
        return None, None


def stall_attribution(kernel):
    """Where the wavefronts of `kernel` spend their cycles: the committed SQ_WAVE_CYCLES / SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY /
    SQ_WAIT_ANY pass of this command line (profiles/rNN_stall_counters.json, tools/pmc_diag.sh) -- a recorded constant like `traffic`."""
    d, src = _profile_json("stall_counters.json")
    try:
        k = d["kernels"][kernel]
        return {"active": k["active"], "issue_stall": k["issue_stall"], "waitcnt": k["waitcnt"], "source": src,
                "note": "fractions of the kernel's wave cycles: an instruction executing / ready but the pipe taken (two wavefronts per SIMD: by the "
                        "other one) / parked on s_waitcnt; a committed counter pass, not measured by this run"}
    except (KeyError, TypeError):
        return None


def measured_valu(kernel=None):
    """VALU wave-instructions from the committed rocprofv3 SQ_INSTS_VALU pass of this command line (profiles/rNN_valu_counters.json,
    tools/gpu_suite.sh pmc): per step over every kernel, or per launch of `kernel`'s transaction grid. None when the file is missing."""
    d, src = _profile_json("valu_counters.json")
    try:
        if d is None:
            return None, None
        d = dict(d, source_file=src)
        if kernel is None:
            return d["insts_valu_per_step"], d
        return d["kernels"][kernel]["insts_valu_largest_grid_mean"], d
    except (KeyError, ValueError):
        return None, None


# cycles one SIMD spends on one wave-instruction of this instruction mix when nothing stalls: 61 % v_mad_u64_u32 at 4.6-4.9, the 64-bit
# shifts / adds and v_mul_lo_u32 at 4.1-4.3, a few 32-bit VOP2 at 2.1 (tools/microbench/instbench.hip -> profiles/r03_instbench.txt)
VALU_CYCLES_PER_INST = 4.3


def deep_valu(launch_ms, torch, local):
    """the integer-issue side of k_smt on the deep state: SQ_INSTS_VALU of its transaction launch from the committed counter pass taken with
    --accounts 2^20 (profiles/rNN_valu_counters_deep.json) against THIS run's launch time"""
    d, src = _profile_json("valu_counters_deep.json")
    try:
        insts = float(d["kernels"]["k_smt"]["insts_valu_largest_grid_mean"])
    except (KeyError, TypeError, ValueError):
        return {}
    props = torch.cuda.get_device_properties(local)
    clock_hz = float(getattr(props, "clock_rate", 0) or 2400000) * 1e3
    return {"insts_valu_per_launch": int(insts), "frac_valu": round(insts * VALU_CYCLES_PER_INST / (props.multi_processor_count * 4 * clock_hz * launch_ms * 1e-3), 5),
            "valu_source": src}


def poseidon_valu(kernel):
    """SQ_INSTS_VALU per launch of 2^20 permutations of one poseidon_batch_kernel instantiation (profiles/rNN_poseidon_valu.json), or None"""
    d, _ = _profile_json("poseidon_valu.json")
    try:
        return float(d["kernels"][kernel]["insts_valu_per_launch"])
    except (KeyError, ValueError, TypeError):
        return None


def host_limits():
    """(CPUs this process may really use, bytes of memory it may still take): the cgroup's quota when there is one -- a container
    that shows 256 logical CPUs and 3 TB can be held to 16 CPUs' worth of time and 300 GiB (the GPU boxes of this project are)."""
    cpus = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cpus = max(1, min(cpus, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    try:
        cpus = min(cpus, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        mem = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except (OSError, IndexError):
        mem = 32 << 30
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        if mx != "max":
            mem = min(mem, int(mx) - int(open("/sys/fs/cgroup/memory.current").read()))
    except (OSError, ValueError):
        try:
            mem = min(mem, int(open("/sys/fs/cgroup/memory/memory.limit_in_bytes").read()) - int(open("/sys/fs/cgroup/memory/memory.usage_in_bytes").read()))
        except (OSError, ValueError):
            pass
    return cpus, max(mem, 1 << 30)


def cpu_baseline(n_tx, L, max_l1, F, workers):
    """The CPU oracle (restated CPU path, kind "port") on a bounded sample of the same workload: `workers` processes, one batch
    of `n_tx` transactions each, started together (tests/cpu_baseline_worker.py); value = all their transactions / the slowest run."""
    script = os.path.join(ROOT, "tests", "cpu_baseline_worker.py")
    start = time.time() + 10.0 + 0.014 * n_tx + 0.05 * workers   # every worker has built its batch by then
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
            "sample": "RollupMain(nTx=%d,nLevels=%d,maxL1Tx=%d,maxFeeTx=%d): %d processes x one batch, %.1f s; the host shows %d logical CPUs, "
                      "this process may use %d (cgroup quota / affinity) and %.0f GB of memory"
                      % (n_tx, L, max_l1, F, workers, dt, os.cpu_count() or 0, host_limits()[0], host_limits()[1] / 1e9)}


def node_host_line(args, packed_file, exp_file, Bp, inflight):
    """The same loop driven from Node.js through the N-API addon (tests/node/bench_facade.js): the host the north star names."""
    import shutil
    node = shutil.which("node")
    addon = os.path.join(ROOT, "circuits_amd", "node", "hermez_addon.node")
    if node is None or not os.path.exists(addon):
        return {"error": "node or the addon is not available"}
    try:
        cmd = [node, os.path.join(ROOT, "tests", "node", "bench_facade.js"), packed_file] + [str(x) for x in (args.nTx, args.nLevels, args.maxL1Tx, args.maxFeeTx,
               Bp, inflight, args.steps, args.warmup)] + [exp_file]
        # every HIP call of the Node host from ONE pool thread, in the order a single-threaded host issues them (two pool threads
        # driving one context each measured 54.6 ms per step against 41.9: HZ_NODE_UV_THREADS overrides for experiments)
        env = dict(os.environ, UV_THREADPOOL_SIZE=os.environ.get("HZ_NODE_UV_THREADS", "1"))
        for kv in filter(None, os.environ.get("HZ_NODE_ENV", "").split(",")):   # experiments: extra environment of the Node process
            k, _, v = kv.partition("=")
            env[k] = v
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except (subprocess.TimeoutExpired, ValueError) as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def flagged_point(bp, nctx):
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "12", "--warmup", "3", "--batches-per-launch", str(bp), "--inflight", str(nctx),
           "--distinct-batches", "4", "--cpu-sample", "0", "--no-withdraw", "--no-e2e", "--no-poseidon", "--no-export", "--no-node", "--no-deep-state",
           "--no-sweep", "--no-shard", "--no-verify", "--latency-scheduling"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        ln = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if r.returncode != 0 or not ln:
            return {"contexts": nctx, "error": "child exited with %d: %s" % (r.returncode, (r.stderr or "").strip().splitlines()[-1][:200] if r.stderr else "")}
        d = json.loads(ln[-1])
        return {"contexts": nctx, "ms_per_step": d["ms_per_step"], "tx_per_s": d["value"], "process": "bench.py --batches-per-launch %d --inflight %d --latency-scheduling in a process of its own" % (bp, nctx)}
    except Exception as e:   # noqa: BLE001 -- a secondary figure must never cost the main line
        return {"contexts": nctx, "error": "%s: %s" % (type(e).__name__, e)}


def with_node_host(args):
    """`value_node` needs a process of its own, and needs this one to stay off the GPU: measured (round 3), a Node host started as a
    child of a Python process that still holds its HIP runtime -- idle, contexts freed, but its dozen hardware queues alive -- runs
    the same loop at 1.14 M tx/s instead of 1.61 M (the device's hardware queue slots are oversubscribed and time-sliced). So the
    benchmark proper runs in a worker process (this file again, --gpu-worker), which leaves the packed batches behind; when it has
    exited the Node host replays the upload-inclusive loop on them, and this process prints the one merged line."""
    import shutil
    import tempfile
    keep_dir = tempfile.mkdtemp(prefix="hz_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)   # a directory of our own: no predictable path
    keep = os.path.join(keep_dir, "batches")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--gpu-worker", "--keep-packed", keep]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            sys.stdout.write(r.stdout)
            sys.exit(r.returncode or 1)
        out = json.loads(lines[-1])
        Bp, inflight = out["config"]["batches_per_launch"], out["config"]["contexts_in_flight"]
        if os.path.exists(keep + ".packed"):
            node_line = node_host_line(args, keep + ".packed", keep + ".json", Bp, inflight)
            if "value_node" in node_line:
                out["value_node"] = round(node_line["value_node"], 1)
                if out.get("value_e2e"):
                    node_line["ratio_to_value_e2e"] = round(node_line["value_node"] / out["value_e2e"], 4)
            out["node_host"] = node_line
        if out.get("batches_sweep") and not args.no_sweep:
            # four HZ_FLAG_LATENCY contexts in flight, one / two batches each: a process of its own with the device to itself
            for pt in out["batches_sweep"]:
                if "latency_flag_x2" in pt and pt["latency_flag_x2"] is None:   # two flagged contexts in flight (left to this process by the worker)
                    pt["latency_flag_x2"] = flagged_point(pt["batches_per_launch"], 2)
                if pt.get("batches_per_launch") in (1, 2) and "latency_flag_x2" in pt:
                    pt["latency_flag_x4"] = flagged_point(pt["batches_per_launch"], 4)
        print(json.dumps(out))
    finally:
        shutil.rmtree(keep_dir, ignore_errors=True)


def bench_sharded(args, L, D, packed, expected):
    """BASELINE config 4: ONE batch sharded by transaction index over the ranks (circuits_amd/multigpu.py): one all_gather of the
    160-byte data-availability records, FeeTx + HashInputs on rank 0. Returns the result object (rank 0) or None."""
    torch = D.torch
    from circuits_amd.multigpu import ShardedBatch
    nTx, lv, m1, F = args.nTx, args.nLevels, args.maxL1Tx, args.maxFeeTx
    c = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=D.local)
    c.upload(0, packed)   # the same batch on every rank (same seed)
    stream = torch.cuda.Stream(device=D.local)

    def alloc(n):
        t = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # the fill runs on torch's current stream, the pass on its own non-blocking one: without this the zeros may land on top of the first export
        return t

    def all_gather(recv, send):
        if D.world == 1 and not D.force:
            with torch.cuda.stream(stream):
                recv.copy_(send)
        elif D.backend == "gloo":   # test hook (CPU collectives): through the host, blocking -- "send complete, recv filled" on return
            stream.synchronize()
            h = send.cpu()
            parts = [torch.empty_like(h) for _ in range(D.world)]
            D.dist.all_gather(parts, h)
            recv.copy_(torch.cat(parts))
            torch.cuda.synchronize()
        else:
            with torch.cuda.stream(stream):   # RCCL enqueues on the current stream: ordered with the export / import kernels
                D.dist.all_gather_into_tensor(recv, send)

    def broadcast(buf):
        if D.world == 1 and not D.force:
            return
        if D.backend == "gloo":   # test hook (CPU collectives): through the host, blocking
            stream.synchronize()
            h = buf.cpu()
            D.dist.broadcast(h, src=0)
            if D.rank != 0:
                buf.copy_(h)
            torch.cuda.synchronize()
        else:
            with torch.cuda.stream(stream):
                D.dist.broadcast(buf, src=0)

    sb = ShardedBatch(c, L, nTx, D.rank, D.world, alloc, all_gather, broadcast, force_split=D.force)
    sb.step(stream.cuda_stream)
    if D.rank == 0 and not args.no_verify:
        assert c.get("main.hashGlobalInputs") == expected, "hashGlobalInputs mismatch (sharded)"
    for _ in range(max(1, args.warmup)):
        sb.step(stream.cuda_stream)
    steps = max(4, args.steps)
    dt = D.timed(lambda: [sb.step(stream.cuda_stream) for _ in range(steps)])
    # phases of one pass on this rank, HIP events on the pass's stream (gloo test hook: the collectives block on the host, their time
    # shows as host time between the phases, not on the stream)
    phases = {}
    for _ in range(3):
        marks = []

        def mark(label):
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            marks.append((label, e))
        sb.step(stream.cuda_stream, mark)
        torch.cuda.synchronize()
        for (l0, e0), (l1, e1) in zip(marks, marks[1:]):
            phases[l1] = phases.get(l1, 0.0) + e0.elapsed_time(e1) / 3
    mine = {"rank": D.rank, "transactions": sb.count, "shard_ms": round(phases.get("shard", 0.0), 3), "all_gather_us": round(phases.get("all_gather", 0.0) * 1e3, 1),
            "tail_ms": round(phases.get("tail", 0.0), 3), "broadcast_us": round(phases.get("broadcast", 0.0) * 1e3, 1), "expand_ms": round(phases.get("expand", 0.0), 3)}
    per_rank = [mine]
    if D.world > 1:
        per_rank = [None] * D.world
        D.dist.all_gather_object(per_rank, mine)
    res = None
    if D.rank == 0:
        abytes = algorithmic_bytes_per_tx(lv, F) * nTx
        res = {"value": round(nTx * steps / dt, 1), "unit": "tx-witnesses/s", "scaling": "strong", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
               "rccl_ranks": D.world if (D.backend == "nccl" and (D.world > 1 or D.force)) else 0,
               "backend": D.backend if (D.world > 1 or D.force) else None,
               "per_rank": per_rank,
               "slowest_shard_ms": max(p["shard_ms"] for p in per_rank), "all_gather_us": max(p["all_gather_us"] for p in per_rank),
               "broadcast_us": max(p["broadcast_us"] for p in per_rank),
               "hardware_note": "RCCL between two GPUs over xGMI has not run on any box this code was built on (one-GPU boxes): unmeasured on hardware until "
                                "the driver's 8-GPU run; rccl_ranks = 0 means the collectives of this line did not go through RCCL",
               "parallelism": "tx-shard%d" % D.world,
               "collective": "one all_gather of %d B%s per step (%s)" % (sb.slot * D.world, " + one broadcast of %d B (SHA-256 blocks split over the ranks)" % c.sha_state_bytes()
                                                                          if sb.split_tail else "", D.backend),
               "transactions_per_rank": sb.count,
               "whole_pass_GBs": round(abytes / (dt / steps) / 1e9, 2)}
    del sb, c
    return res


def bench_withdraw(args, L, D, launches=None, steps=None):
    """BASELINE config 5: 2^20 independent Withdraw(nLevels) witnesses (SMTVerifier + HashState + 2-block SHA-256 bit witness),
    `--withdraw-per-launch` instances per kernel launch, exits drawn from one exit tree of `--withdraw-leaves` leaves (hashed on
    the device). One lane per witness. Returns the result object (rank 0)."""
    torch = D.torch
    from circuits_amd import builder as B
    lv = args.nLevels
    n_leaves = args.withdraw_leaves
    fx = B.ExitTreeFixture(n_leaves, device=D.local)
    idxs = sorted(fx.exit_leaves)
    N = args.withdraw_per_launch
    free_b, _ = torch.cuda.mem_get_info()
    probe = L.ctx("withdraw", nLevels=lv, device=D.local, n_instances=1)
    wl = probe.witness_len()
    names = [n for n, _ in probe.input_names()]
    del probe
    N = max(64, min(N, int((free_b - (6 << 30)) // (wl * 32 * 1.02)) // 64 * 64))
    c = L.ctx("withdraw", nLevels=lv, device=D.local, n_instances=N)
    # every instance withdraws a different leaf when the tree has that many (instance k -> leaf k * 40503 mod n_leaves: scattered)
    pick = [idxs[(k * 40503) % n_leaves] for k in range(N)]
    uniq = {}
    for i in set(pick):
        uniq[i] = B.withdraw_input(fx, i, lv)
    for name in names:
        c.set_input(name, [uniq[i][0][name] for i in pick], instance=-1)
    stream = torch.cuda.Stream(device=D.local)
    c.enqueue(stream.cuda_stream)
    c.check()
    for k in (0, N // 2 + 1, N - 1):
        assert c.get("main.hashGlobalInputs", k) == uniq[pick[k]][1], "hashGlobalInputs mismatch (withdraw)"
    launches = launches or max(1, (args.withdraw_total + N - 1) // N)
    steps = steps or args.steps
    for _ in range(max(1, args.warmup)):
        c.enqueue(stream.cuda_stream)
        c.check()

    def run():
        for _ in range(steps):
            for _ in range(launches):
                c.enqueue(stream.cuda_stream)
            c.check()

    dt = D.timed(run)
    # each kernel alone on the device (HIP events on its stream): the per-kernel rooflines
    c.set_profiling(True, exclusive=True)
    acc = {}
    for _ in range(3):
        c.enqueue(stream.cuda_stream)
        c.check()
        for name, ms, by, units in c.profile():
            a = acc.setdefault(name, [0.0, by])
            a[0] += ms / 3
    c.set_profiling(False)
    res = None
    if D.rank == 0:
        total = launches * N * steps * D.world
        abytes = wl * 32 * N
        ms_launch = dt / (steps * launches) * 1e3   # the two kernels of a launch run concurrently: wall time per launch
        dk = max(acc, key=lambda k: acc[k][1])
        res = {"metric": "withdraw witnesses/sec (nLevels=%d)" % lv, "value": round(total / dt, 1), "unit": "witnesses/s", "n_gpus": D.world,
               "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3), "scaling": "weak",
               "config": {"workload": "withdraw nLevels=%d, %d witnesses per step in %d launches of %d" % (lv, launches * N, launches, N),
                          "witness_elements": wl, "exit_tree_leaves": n_leaves, "distinct_leaves_per_launch": len(uniq)},
               "whole_launch": {"achieved_GBs": round(abytes / (ms_launch * 1e-3) / 1e9, 2), "frac": round(abytes / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                "launch_ms": round(ms_launch, 3), "algorithmic_bytes_per_launch": int(abytes), "note": "k_withdraw_sha || k_withdraw, wall time"},
               "roofline": {"bound": "hbm", "kernel": "k_" + dk, "achieved": round(acc[dk][1] / (acc[dk][0] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(acc[dk][1] / (acc[dk][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": measured_traffic("k_" + dk, None)[0], "traffic_source": measured_traffic("k_" + dk, None)[1],
                            "launch_ms": round(acc[dk][0], 3), "algorithmic_bytes_per_launch": int(acc[dk][1])},
               "kernels_ms": {k: round(v[0], 3) for k, v in acc.items()},
               "kernels_GBs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in acc.items() if v[0] > 0}}
    del c
    return res


def bench_export(args, L, D, ctxs, streams, run_enqueue, nTx, Bp):
    """The witness in a prover's order (SURVEY 8a' K8, 8d "D2H excluded / included reported separately"): hz_witness_export_dev turns
    one instance of the signal-major buffer into w[var] on the device. The map is this layout's stored signals in COMPONENT-MAJOR order
    (hz_component_major_index: every transaction's signals together, the shape of a constraint-reducing circom compile's numbering; the
    compiler itself cannot run here). Returns the `export` object of the line."""
    import tempfile
    torch = D.torch
    c0 = ctxs[0]
    wl = c0.witness_len()
    t0 = time.time()
    index = c0.component_major_index()
    maps = [c.symmap_from_index(index) for c in ctxs]
    tables = [m.upload() for m in maps]
    t_plan = time.time() - t0
    nbuf = 2
    outs = [torch.empty(wl * 32, dtype=torch.uint8, device="cuda") for _ in range(nbuf)]
    torch.cuda.synchronize()
    s0 = streams[0]
    # (1) one batch, the device otherwise idle: HIP events on the export's stream
    times = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s0)
        maps[0].export_dev(outs[0].data_ptr(), r % Bp, stream=s0.cuda_stream)
        e1.record(s0)
        torch.cuda.synchronize()
        if r:
            times.append(e0.elapsed_time(e1))
    exp_ms = sorted(times)[len(times) // 2]
    # the exported vector holds the public output where the map says (parity of the whole vector: tests/test_export_dev.py)
    pos = int((index == c0.lookup("main.hashGlobalInputs")).nonzero()[0][0])
    got = int.from_bytes(bytes(outs[0][32 * pos:32 * pos + 32].cpu().numpy()), "little")
    assert got == c0.get("main.hashGlobalInputs", 5 % Bp), "exported hashGlobalInputs mismatch"
    # (2) the step WITH the export of every batch it produced (each context's exports follow its kernels on its stream)
    inflight = len(ctxs)

    def run(n):
        pend = [False] * inflight
        for i in range(n):
            k = i % inflight
            if pend[k]:
                ctxs[k].check()
            run_enqueue(k)
            for b in range(Bp):
                maps[k].export_dev(outs[(k + b) % nbuf].data_ptr(), b, stream=streams[k].cuda_stream)
            pend[k] = True
        for k in range(inflight):
            if pend[k]:
                ctxs[k].check()
    esteps = max(inflight, min(args.steps, 4))
    run(inflight)
    dt = D.timed(lambda: run(esteps))
    # the same with FOUR batches per export call (hz_witness_export_range_dev: the sections whose unit is the instance -- HashInputs -- are
    # then read as whole 128-byte lines instead of 32 bytes out of every KB), one ring of four vectors per context when the memory is there
    group, dt4, exp4_ms = 4, None, None
    del outs
    torch.cuda.empty_cache()
    if Bp % group == 0:
        try:
            rings = [torch.empty(group * wl * 32, dtype=torch.uint8, device="cuda") for _ in ctxs]
            torch.cuda.synchronize()
        except RuntimeError:
            rings = None
        if rings:
            def run4(n):
                pend = [False] * inflight
                for i in range(n):
                    k = i % inflight
                    if pend[k]:
                        ctxs[k].check()
                    run_enqueue(k)
                    for b in range(0, Bp, group):
                        L._check(L.c.hz_witness_export_range_dev(ctxs[k].h, maps[k].h, b, group, rings[k].data_ptr(), streams[k].cuda_stream))
                    pend[k] = True
                for k in range(inflight):
                    if pend[k]:
                        ctxs[k].check()
            tt = []
            for r in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(s0)
                L._check(L.c.hz_witness_export_range_dev(c0.h, maps[0].h, (4 * r) % Bp, group, rings[0].data_ptr(), s0.cuda_stream))
                e1.record(s0)
                torch.cuda.synchronize()
                if r:
                    tt.append(e0.elapsed_time(e1) / group)
            exp4_ms = sorted(tt)[len(tt) // 2]
            run4(inflight)
            dt4 = D.timed(lambda: run4(esteps))
            del rings
            torch.cuda.empty_cache()
    # (3) delivered to the host: export + device-to-host copy into pinned memory, a few batches (PCIe bound)
    nh = 3
    pin = L.host_alloc(wl * 32)
    maps[0].export_host(instance=0, out=pin)
    t1 = time.perf_counter()
    for b in range(nh):
        maps[0].export_host(instance=(b + 1) % Bp, out=pin)
    dt_host = (time.perf_counter() - t1) / nh
    L.host_free(pin)
    # (4) one batch as a snarkjs .wtns in that order (the ring of two pinned buffers feeding fwrite), to a memory-backed file
    d = "/dev/shm" if os.path.isdir("/dev/shm") else None
    fd, path = tempfile.mkstemp(suffix=".wtns", dir=d)
    os.close(fd)
    try:
        t2 = time.perf_counter()
        maps[0].write_wtns(path, instance=0)
        wtns_s = time.perf_counter() - t2
        wtns_bytes = os.path.getsize(path)
    finally:
        os.unlink(path)
    res = {"order": "component-major (hz_component_major_index): the stored signals, every transaction's together -- the shape of a reducing compile's numbering",
           "variables_per_batch": int(wl), "bytes_per_batch": int(wl * 32),
           "export_ms_per_batch": round(exp_ms, 3), "export_GBs_read_plus_write": round(2 * wl * 32 / exp_ms / 1e6, 1),
           "frac_of_hbm_peak": round(2 * wl * 32 / exp_ms / 1e6 / HBM_PEAK_GBS, 4),
           "value_export": round(nTx * Bp * esteps * D.world / dt, 1), "value_export_ms_per_step": round(dt / esteps * 1e3, 3), "value_export_steps": esteps,
           "export_ms_per_batch_4_per_call": round(exp4_ms, 3) if exp4_ms else None,
           "value_export_4_per_call": round(nTx * Bp * esteps * D.world / dt4, 1) if dt4 else None,
           "value_export_4_per_call_ms_per_step": round(dt4 / esteps * 1e3, 3) if dt4 else None,
           "delivered_host_ms_per_batch": round(dt_host * 1e3, 2), "delivered_host_GBs": round(wl * 32 / dt_host / 1e9, 2),
           "value_delivered_host": round(nTx / dt_host, 1), "pcie_ceiling_tx_per_s": round(nTx / (wl * 32 / 63e9), 1),
           "wtns_write_s": round(wtns_s, 2), "wtns_bytes": int(wtns_bytes), "wtns_GBs": round(wtns_bytes / wtns_s / 1e9, 2),
           "plan_build_s": round(t_plan / len(ctxs), 2), "plan_device_bytes": int(tables[0]),
           "note": "export_ms_per_batch: hz_witness_export_dev of ONE batch on the otherwise idle device, HIP events on its stream (target of the round-4 review: <= 2.5 ms); "
                   "*_4_per_call: four batches per hz_witness_export_range_dev call (HashInputs' unit is the batch: four of them make whole 128-byte lines). "
                   "value_export: the timed step plus the export of every batch it produced into a ring of device buffers -- a full copy of 3.86 GB per batch costs more "
                   "than computing it (the step writes each byte once, the export reads and writes it again); a consumer on the same GPU can take the map's "
                   "indirection instead (hz_symmap_dev_index). value_delivered_host: export + one device-to-host copy into pinned memory, PCIe Gen5 x16 <= 63 GB/s: "
                   "the ceiling is pcie_ceiling_tx_per_s. wtns_write_s: one batch as .wtns through the pinned ring into a memory-backed file"}
    del maps
    return res


def bench_shard_standin(L, D, nTx, lv, m1, F, packed_ptr, pbytes, expected, world=8):
    """A stand-in for BASELINE config 4 on ONE GPU: the batch cut into `world` contiguous transaction ranges exactly
    as bench.py --gpus 8 --shard-tx cuts it (circuits_amd/multigpu.py), every range evaluated ALONE on this device -- what its own GPU
    would see -- then rank 0's tail and every rank's share of the SHA-256 blocks, HIP events on the pass's stream. The two collectives
    (328 KB all_gather, 73 KB broadcast) are not in the figure: RCCL over xGMI has not run on any box this code was built on."""
    torch = D.torch
    c = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=D.local, flags=2)   # one batch per GPU: latency scheduling
    stream = torch.cuda.Stream(device=D.local)
    c.upload(0, packed_ptr, pbytes, stream.cuda_stream)
    ranges = [L.shard_range(nTx, world, r) for r in range(world)]
    rec = c.da_record_bytes()
    slot = max(n for _, n in ranges) * rec
    recv = torch.zeros(slot * world, dtype=torch.uint8, device="cuda")
    sha = torch.zeros(c.sha_state_bytes(), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        fn()
        e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    reps = 3
    shard_ms, expand_ms, tail_ms = [0.0] * world, [0.0] * world, 0.0
    for rep in range(reps + 1):
        for r in range(world - 1, -1, -1):   # rank 0 last: its context then holds its own shard's scratch for the tail
            f, n = ranges[r]
            c.set_shard(f, n, r == 0)

            def shard(r=r):
                c.enqueue(stream.cuda_stream)
                c.da_export(recv.data_ptr() + r * slot, stream.cuda_stream)
            t = timed(shard)
            if rep:
                shard_ms[r] += t / reps

        def tail():
            for r in range(1, world):
                f, n = ranges[r]
                c.da_import(f, n, recv.data_ptr() + r * slot, stream.cuda_stream)
            c.enqueue_tail_chain(stream.cuda_stream)
            c.sha_export(sha.data_ptr(), stream.cuda_stream)
        t = timed(tail)
        if rep:
            tail_ms += t / reps
        for r in range(world):
            f, n = L.shard_range(c.sha_blocks(), world, r)
            t = timed(lambda f=f, n=n: c.sha_expand(f, n, None, stream.cuda_stream))
            if rep:
                expand_ms[r] += t / reps
        c.check()
        assert c.get("main.hashGlobalInputs") == expected, "hashGlobalInputs mismatch (shard stand-in)"
    total = max(shard_ms) + tail_ms + max(expand_ms)
    del c
    return {"world": world, "transactions_per_rank": [n for _, n in ranges], "shard_ms": [round(x, 3) for x in shard_ms], "slowest_shard_ms": round(max(shard_ms), 3),
            "tail_ms_rank0": round(tail_ms, 3), "expand_ms": [round(x, 3) for x in expand_ms], "modelled_ms_per_batch": round(total, 3),
            "modelled_tx_per_s": round(nTx / total * 1e3, 1),
            "note": "config 4 as named (one batch over 8 GPUs), modelled on one GPU: slowest shard alone + rank 0's FeeTx / message / SHA-256 chain + slowest share of the "
                    "block expansion; the all_gather and the broadcast (latency-sized) are NOT included -- RCCL between GPUs is unmeasured on hardware. A 256-transaction "
                    "shard is 4 wavefronts per kernel: the dependent chains (33 level hashes, 148 ladder steps) set its time, so 8 GPUs do not make one batch 8 x faster"}


class Rotation:
    """The measured loop. The packed inputs of `n` distinct batches lie TWICE in a row in a source buffer (pinned host memory for the
    upload-inclusive figures, HBM for `value`: "inputs resident when the timed region starts"), so that any window of consecutive
    batches modulo n is contiguous. Context k's instance b holds batch (k * bp + b + r) mod n in round r, and EVERY step is a new round:
    right after a step's enqueue the next round's window is staged (hz_inputs_stage_range: one copy beside the step's kernels), the next
    enqueue scatters it into the witness layout first. No instance sees the same batch in two consecutive steps, so nothing that a step
    finds in the persistent witness buffer from the step before (csrc/ctx.hip "constant marks") is its own batch's."""

    def __init__(self, ctxs, streams, bp, n, pbytes, expected, host_src, dev_src=None):
        self.cs, self.st, self.bp, self.n, self.pb, self.exp = ctxs, streams, bp, n, pbytes, expected
        self.host, self.dev = host_src, dev_src
        self.cur = [0] * len(ctxs)   # the round whose inputs context k's last enqueue consumed (or that were uploaded)
        self.nxt = [0] * len(ctxs)   # the round staged for its next enqueue (== cur: nothing staged)

    def slot(self, k, b, r=None):
        return (k * self.bp + b + (self.cur[k] if r is None else r)) % self.n

    def load(self):   # round 0 through hz_inputs_upload (copy + scatter)
        for k, c in enumerate(self.cs):
            for b in range(self.bp):
                c.upload(b, self.host + self.slot(k, b, 0) * self.pb, self.pb, self.st[k].cuda_stream)
            self.cur[k] = self.nxt[k] = 0

    def stage(self, k, src):
        r = self.nxt[k] + 1
        b = 0
        while b < self.bp:   # one call when bp <= n
            run = min(self.bp - b, self.n)
            self.cs[k].stage_range(b, run, src + self.slot(k, b, r) * self.pb, self.pb)
            b += run
        self.nxt[k] = r

    def enqueue(self, k):
        self.cs[k].enqueue(self.st[k].cuda_stream)
        self.cur[k] = self.nxt[k]

    def step(self, k, src):
        self.enqueue(k)
        if src:
            self.stage(k, src)

    def go(self, n, src):
        """n steps round-robin over the contexts in flight; src: where the next round's inputs come from (None: no rotation, every
        step re-evaluates the inputs it already holds -- experiments only)"""
        pend = [False] * len(self.cs)
        for i in range(n):
            k = i % len(self.cs)
            if pend[k]:
                self.cs[k].check()
            self.step(k, src)
            pend[k] = True
        for k in range(len(self.cs)):
            if pend[k]:
                self.cs[k].check()

    def drain(self):   # consume what is still staged
        for k in range(len(self.cs)):
            if self.nxt[k] != self.cur[k]:
                self.enqueue(k)
                self.cs[k].check()

    def verify(self, what, instances=None):
        for k, c in enumerate(self.cs):
            for b in (range(self.bp) if instances is None else instances):
                assert c.get("main.hashGlobalInputs", b) == self.exp[self.slot(k, b)], "hashGlobalInputs mismatch (%s, context %d, batch %d, round %d)" % (what, k, b, self.cur[k])


def doubled_sources(L, torch, src_ptr, n, pbytes):
    """(pinned, device tensor): the n packed batches at src_ptr twice in a row, in pinned host memory and in HBM"""
    pin2 = L.host_alloc(2 * n * pbytes)
    ctypes.memmove(pin2, src_ptr, n * pbytes)
    ctypes.memmove(pin2 + n * pbytes, src_ptr, n * pbytes)
    host_view = torch.frombuffer((ctypes.c_char * (2 * n * pbytes)).from_address(pin2), dtype=torch.uint8)
    dev = torch.empty(2 * n * pbytes, dtype=torch.uint8, device="cuda")
    dev.copy_(host_view)
    torch.cuda.synchronize()
    return pin2, dev


def respawn(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one process per GPU) and hand over."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


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
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU-baseline processes (0 = every CPU this process may use -- cgroup quota, affinity -- as far as a third of its memory allows: a RollupMain(2048, 32, ..) oracle holds a 3.9 GB witness)")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="nTx of the CPU-baseline sample (0 = skip; default: the headline shape, one batch per process)")
    ap.add_argument("--no-sweep", action="store_true", help="skip single_batch_latency_ms and batches_sweep (the occupancy points of SURVEY 8d)")
    ap.add_argument("--no-export", action="store_true", help="skip the export figures (the witness in variable order on the device / delivered to the host / as .wtns)")
    ap.add_argument("--no-deep-state", action="store_true", help="skip the deep_state line (the same step on a state of 2^20 accounts)")
    ap.add_argument("--deep-accounts-log2", type=int, default=20)
    ap.add_argument("--batches-per-launch", type=int, default=32,
                    help="independent batches evaluated by ONE set of kernel launches (context with n_instances = B): more wavefronts per launch")
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="differently seeded batches built for the run (0 = one per resident instance: batches-per-launch x inflight); fewer are reused round-robin")
    ap.add_argument("--build-workers", type=int, default=0, help="host processes building the synthetic batches (0 = min(batches, CPUs - 2, 64))")
    ap.add_argument("--workload", choices=["rollup-main", "withdraw"], default="rollup-main",
                    help="withdraw = BASELINE config 5 (2^20 independent Withdraw(nLevels) witnesses) as the main line")
    ap.add_argument("--withdraw-total", type=int, default=1 << 20)
    ap.add_argument("--withdraw-per-launch", type=int, default=1 << 16)
    ap.add_argument("--withdraw-leaves", type=int, default=1 << 16, help="leaves of the exit tree the withdrawals are drawn from (SURVEY 8d: 2^16)")
    ap.add_argument("--latency-scheduling", action="store_true",
                    help="HZ_FLAG_LATENCY contexts (their concurrent kernel chains on CU-masked streams of their own): the latency regime, "
                         "one to four batches per launch; single-batch latency: --batches-per-launch 1 --inflight 1 --latency-scheduling, "
                         "its throughput: --batches-per-launch 1 --inflight 4 --latency-scheduling")
    ap.add_argument("--solo", action="store_true", help="with --latency-scheduling --inflight 1: HZ_FLAG_SOLO as well (nothing else on the device: the SMT chain kernel in its latency form)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-rotate", action="store_true", help="experiments: every step re-evaluates the inputs its instances already hold (rounds 1-5's `value` loop) "
                                                              "instead of a new batch per instance and step")
    ap.add_argument("--no-poseidon", action="store_true", help="skip the Poseidon-BN254/sec secondary metric")
    ap.add_argument("--no-withdraw", action="store_true", help="skip the config-5 (withdraw) secondary line")
    ap.add_argument("--no-e2e", action="store_true", help="skip the upload-inclusive run (value_e2e)")
    ap.add_argument("--python-builder", action="store_true", help="build the synthetic batches with the Python builder in a process pool (host hashing) instead of the native builder + hz_poseidon_dag")
    ap.add_argument("--no-node", action="store_true", help="skip value_node (the same loop driven from Node.js through the N-API addon)")
    ap.add_argument("--keep-packed", default="", help="leave the packed batches and their expected outputs at PATH.packed / PATH.json (the Node host line reads them)")
    ap.add_argument("--gpu-worker", action="store_true", help=argparse.SUPPRESS)   # internal: the process that touches the GPU (see main)
    ap.add_argument("--no-shard", action="store_true", help="N > 1: skip the tx-sharded (config 4, strong scaling) secondary line")
    ap.add_argument("--calibrate-copy", action="store_true",
                    help="profiling aid: one 1 GiB device-to-device tensor copy before the timed region, a known byte count that "
                         "calibrates the FETCH_SIZE / WRITE_SIZE counters of a rocprofv3 --pmc pass (tools/profile.sh)")
    ap.add_argument("--shard-timeout", type=int, default=240, help="N > 1: seconds the tx-sharded secondary line may take before it is given up")
    ap.add_argument("--shard-tx", action="store_true", help="only the tx-sharded line (BASELINE config 4) as the main line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)
    if (args.gpus <= 1 and "WORLD_SIZE" not in os.environ and not args.gpu_worker and not args.no_node and not args.no_e2e
            and args.workload == "rollup-main" and not args.shard_tx):
        return with_node_host(args)
    D = Dist()
    torch = D.torch
    rank, world, local = D.rank, D.world, D.local
    if world != max(1, args.gpus) and rank == 0:
        print("bench: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)

    from circuits_amd import lib
    from circuits_amd.batchgen import build_packed_batches
    L = lib()
    if L.device_count() <= 0:
        raise SystemExit("no gfx950 device")
    if args.workload == "withdraw":
        res = bench_withdraw(args, L, D)
        if rank == 0:
            res.update({"warmup": args.warmup, "higher_is_better": True, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic"})
            print(json.dumps(res))
        return D.close()
    nTx, lv, m1, F = args.nTx, args.nLevels, args.maxL1Tx, args.maxFeeTx
    n_acc = args.accounts if args.accounts > 0 else 4 * nTx
    Bp = max(1, args.batches_per_launch)
    # more than 4 contexts (3 streams each) exhaust the runtime's per-queue scratch reservations (HSA_STATUS_ERROR_OUT_OF_RESOURCES)
    inflight = max(1, min(args.inflight, 4, args.steps if args.steps > 0 else 1))
    # every batch keeps its whole witness resident (3.86 GB at the default shape): fit batches x contexts into free HBM
    probe = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=local, n_instances=1)
    layout = probe.packed_layout()
    del probe
    # what one resident batch costs: the library's own count for a context of Bp instances (witness, scratch, upload staging, and up to
    # 16 384 transactions per launch the signature ladder's side buffer -- 84.7 KB per transaction) plus allocator slack

    def per_batch_bytes(b):
        return L.template_device_bytes("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, n_instances=b) / b * 1.02 + (64 << 20)
    per_batch = per_batch_bytes(Bp)
    free_b, _total_b = torch.cuda.mem_get_info()
    if "HZ_BENCH_DEVICE" in os.environ:
        free_b //= world   # test hook: the ranks share one device
    fit = int((free_b - (20 << 30)) // (per_batch * inflight))   # (the export figures take two vectors of one batch and the maps' tables)
    if fit < Bp:
        print("bench: %d batches x %d contexts do not fit %.0f GB of free HBM, using %d batches per launch" % (Bp, inflight, free_b / 1e9, max(1, fit)), file=sys.stderr)
        Bp = max(1, fit)
    # synthetic batches (reference tools/generate-input.js recipe, SURVEY 8d: a state of 4*nTx accounts): one per resident instance,
    # each from its own seed (own accounts, balances, transactions, signatures, Merkle paths). The state size matters: the proofs
    # of a tree of 2^13 leaves reach their leaf at level 13-14, the levels below hash empty subtrees.
    n_res = Bp * inflight
    n_distinct = min(n_res, args.distinct_batches) if args.distinct_batches > 0 else n_res
    t_build = time.time()
    if args.shard_tx:
        n_distinct = 1
    # the sharded line needs ONE batch that every rank holds (seed SEED); the weak-scaling batches are per rank
    want_shard = args.shard_tx or (world > 1 and not args.no_shard)
    seeds = [] if args.shard_tx else [SEED + 1 + 1000 * rank + i for i in range(n_distinct)]
    # the ranks of one node build their batches at the same time: each takes its share of the host cores
    n_build = len(seeds) + (1 if want_shard else 0)
    build_workers = args.build_workers or max(1, min(n_build, 64, ((os.cpu_count() or 2) - 2) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))),
                                                   int(host_limits()[1] // 4 // (3 << 30)) or 1))
    pbytes = layout[0]
    all_seeds = seeds + ([SEED] if want_shard else [])
    native = not args.python_builder and n_acc >= 16 and n_acc & (n_acc - 1) == 0
    builder_stats = None
    if native:
        # the native batch builder (libhz_host.so): walk + signing in C++, each batch's hashes as one DAG on this rank's GPU, the packed
        # inputs written straight into the pinned buffer the uploads read
        from circuits_amd.batchgen import build_packed_batches_native
        pin_all = L.host_alloc(pbytes * len(all_seeds))
        batches, builder_stats = build_packed_batches_native(all_seeds, nTx, lv, m1, F, n_acc, layout, L, local, pin_all)
        batches = [(pin_all + i * pbytes, b[1], b[2]) for i, b in enumerate(batches)]
    else:
        batches = build_packed_batches(all_seeds, nTx, lv, m1, F, n_acc, layout, build_workers)
    t_build = time.time() - t_build
    shared = batches.pop() if want_shard else None
    if shared is not None and native:
        shared = (ctypes.string_at(shared[0], pbytes), shared[1], shared[2])
    n_l2 = (batches[0] if batches else shared)[2]
    if args.shard_tx:
        res = bench_sharded(args, L, D, shared[0], shared[1])
        if rank == 0:
            print(json.dumps({"metric": "rollup-main tx-witnesses/sec (nTx=%d, nLevels=%d)" % (nTx, lv), "value": res["value"], "unit": "tx-witnesses/s",
                              "n_gpus": world, "steps": res["steps"], "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                              "config": {"workload": "rollup-main nTx=%d nLevels=%d maxL1Tx=%d maxFeeTx=%d" % (nTx, lv, m1, F), "parallelism": res["parallelism"],
                                         "collective": res["collective"], "l1_txs": nTx - n_l2, "l2_signed_txs": n_l2},
                              "roofline": {"bound": "hbm", "kernel": "whole sharded pass", "achieved": res["whole_pass_GBs"], "peak": HBM_PEAK_GBS * world,
                                           "unit": "GB/s", "frac": round(res["whole_pass_GBs"] / (HBM_PEAK_GBS * world), 5), "traffic": None}}))
        return D.close()
    # pinned host copies of the packed inputs: the source of every upload
    if native:
        pin = pin_all
    else:
        pin = L.host_alloc(pbytes * n_distinct)
        for i, (pk, _, _) in enumerate(batches):
            ctypes.memmove(pin + i * pbytes, pk, pbytes)
    expected = [b[1] for b in batches]
    del batches
    packed_file = None
    if rank == 0 and world == 1 and args.keep_packed:
        with open(args.keep_packed + ".packed", "wb") as f:
            f.write(ctypes.string_at(pin, pbytes * n_distinct))
        json.dump([str(e) for e in expected], open(args.keep_packed + ".json", "w"))
    ctxs, streams = [], []
    for k in range(inflight):
        c = L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=local, n_instances=Bp,
                  flags=(6 if (args.solo and inflight == 1) else 2) if args.latency_scheduling else 0)
        ctxs.append(c)
        streams.append(torch.cuda.Stream(device=local))
    if os.environ.get("HZ_BENCH_OWN_STREAMS") == "1":   # experiment: the contexts' own streams (what a host without torch passes: NULL)
        class _Null:
            cuda_stream = None
        streams = [_Null() for _ in streams]

    pin2, dsrc_t = doubled_sources(L, torch, pin, n_distinct, pbytes)
    dsrc = None if args.no_rotate else dsrc_t.data_ptr()
    R = Rotation(ctxs, streams, Bp, n_distinct, pbytes, expected, pin2, dsrc)
    R.load()   # inputs resident in HBM before the timed region (round 0 in the witness layout, every batch in packed form)
    # one checked pass per context: parity of every batch's public output with the builder's independently computed value
    ctxs[0].set_profiling(True)
    for k in range(inflight):
        R.enqueue(k)
        ctxs[k].check()
    if not args.no_verify:
        R.verify("first pass")
    full_bytes = {name: by for name, _ms, by, _u in ctxs[0].profile()}   # a fresh buffer holds nothing: every signal was stored
    ctxs[0].set_profiling(False)

    def upload(k):
        for b in range(Bp):
            ctxs[k].upload(b, pin2 + R.slot(k, b) * pbytes, pbytes, streams[k].cuda_stream)

    if args.calibrate_copy:
        a = torch.zeros(1 << 28, dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        del a, b
    # `value`: every step = scatter of the round's packed inputs (HBM -> witness layout) + witness kernels + check, then the next
    # round's window staged from the packed copies in HBM. The first enqueue of the warm-up consumes nothing new; from then on every
    # step of every context evaluates batches its instances did not hold the step before.
    for k in range(inflight):
        if dsrc:
            R.stage(k, dsrc)
    R.go(max(args.warmup, inflight if dsrc else 0), dsrc)
    dt = D.timed(lambda: R.go(args.steps, dsrc))
    R.drain()
    if not args.no_verify:
        R.verify("after the timed region", instances=(0, Bp // 2, Bp - 1))
    dt_e2e = None
    if not args.no_e2e:
        for k in range(inflight):
            R.stage(k, pin2)
        R.go(max(inflight, args.warmup), pin2)   # every timed step below consumes inputs staged during the step before it
        dt_e2e = D.timed(lambda: R.go(args.steps, pin2))
        R.drain()
        if not args.no_verify:
            R.verify("after the upload-inclusive region", instances=(0, Bp - 1))

    # latency of one step alone on the device (wall clock around enqueue + check; `batches_per_launch` batches)
    lat = []
    for _ in range(3):
        if dsrc:
            R.stage(0, dsrc)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        R.enqueue(0)
        ctxs[0].check()
        lat.append((time.perf_counter() - t1) * 1e3)
    single_ms = min(lat)
    # per-kernel device times (HIP events on the stream each kernel is launched on), each kernel alone
    # on the device (in the timed region above the EdDSA and fee-tx chains overlap the hash/SMT chain)
    ctxs[0].set_profiling(True, exclusive=True)  # each kernel alone on the device: durations for the per-kernel roofline
    acc = {}
    reps = 3
    for _ in range(reps):   # (each repetition a new round as well: the chain kernel's time and bytes are those of a step that meets new batches)
        if dsrc:
            R.stage(0, dsrc)
            torch.cuda.synchronize()
        R.enqueue(0)
        ctxs[0].check()
        for name, ms, by, units in ctxs[0].profile():
            a = acc.setdefault(name, [0.0, 0.0, units, 0])
            a[0] += ms / reps      # a kernel launched in several pieces adds up over its launches
            a[1] += by / reps      # algorithmic bytes of THIS launch (k_smt: without what the constant marks let it leave in place)
            a[3] += 1.0 / reps
    ctxs[0].set_profiling(False)
    # upload path alone: one instance's packed inputs, host -> device -> witness layout (HIP events on the stream)
    if streams[0].cuda_stream is None:
        upload_ms = 0.0
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(streams[0]):
            e0.record()
            upload(0)
            e1.record()
        torch.cuda.synchronize()
        upload_ms = e0.elapsed_time(e1) / Bp
    witness_bytes = ctxs[0].witness_len() * 32
    export = None
    if world == 1 and not args.no_export:
        try:
            export = bench_export(args, L, D, ctxs, streams, lambda k: R.step(k, dsrc), nTx, Bp)
        except Exception as e:   # noqa: BLE001 -- a secondary figure must never cost the main line
            export = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    standin = None
    if world == 1 and not args.no_sweep:
        try:
            standin = bench_shard_standin(L, D, nTx, lv, m1, F, pin, pbytes, expected[0])
        except Exception as e:   # noqa: BLE001
            standin = {"error": "%s: %s" % (type(e).__name__, e)}
    deep = None
    if world == 1 and not args.no_deep_state and n_acc < (1 << args.deep_accounts_log2):
        # The same step on a DEEP state: 2^20 accounts instead of 4 * nTx. The proofs then reach their leaves at level ~21 instead
        # of ~14: seven more levels of every chain hash data instead of storing the empty-subtree block -- the sparsity of the
        # reference recipe's state is what the headline rides on. The pre-state is shared (one DenseState
        # hashed on the device, builder.py), every batch brings its own transactions; 8 different batches fill the resident
        # instances round-robin.
        import tempfile
        from circuits_amd import builder as B
        t_deep = time.time()
        kk = args.deep_accounts_log2
        base = B.DenseState.build(kk, seed=SEED ^ 0xD33F, hash_rows=lambda t, n, data: L.poseidon_batch_bytes(t, n, data, device=local))
        t_base = time.time() - t_deep
        n_deep = min(8, n_res)
        dseeds = [SEED + 77000 + i for i in range(n_deep)]
        dpin = L.host_alloc(pbytes * n_deep)
        if native:
            dbatches, _ = build_packed_batches_native(dseeds, nTx, lv, m1, F, 0, layout, L, local, dpin, base=base)
        else:
            fd, base_path = tempfile.mkstemp(suffix=".npz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
            os.close(fd)
            base.save(base_path)
            try:
                dbatches = build_packed_batches(dseeds, nTx, lv, m1, F, 0, layout, min(n_deep, build_workers), base_path=base_path)
            finally:
                os.unlink(base_path)
            for i, (pk, _, _) in enumerate(dbatches):
                ctypes.memmove(dpin + i * pbytes, pk, pbytes)
        del base
        t_deep = time.time() - t_deep
        dexp = [b[1] for b in dbatches]
        del dbatches
        dpin2, ddev_t = doubled_sources(L, torch, dpin, n_deep, pbytes)
        L.host_free(dpin)
        ddev = None if args.no_rotate else ddev_t.data_ptr()
        RD = Rotation(ctxs, streams, Bp, n_deep, pbytes, dexp, dpin2, ddev)
        RD.load()
        for k in range(inflight):
            RD.enqueue(k)
            ctxs[k].check()
        RD.verify("deep state")
        dsteps = max(4, args.steps // 2)
        for k in range(inflight):
            if ddev:
                RD.stage(k, ddev)
        RD.go(max(1, args.warmup, inflight), ddev)
        ddt = D.timed(lambda: RD.go(dsteps, ddev))
        RD.drain()
        RD.verify("deep state, after the timed region", instances=(0, Bp - 1))
        ctxs[0].set_profiling(True, exclusive=True)
        dacc = {}
        for _ in range(2):
            if ddev:
                RD.stage(0, ddev)
                torch.cuda.synchronize()
            RD.enqueue(0)
            ctxs[0].check()
            for name, ms, by, units in ctxs[0].profile():
                a = dacc.setdefault(name, [0.0, 0.0])
                a[0] += ms / 2
                a[1] += by / 2
        ctxs[0].set_profiling(False)
        del ddev_t
        L.host_free(dpin2)
        deep = {"state_accounts": 1 << kk, "value": round(nTx * Bp * dsteps / ddt, 1), "unit": "tx-witnesses/s", "steps": dsteps, "ms_per_step": round(ddt / dsteps * 1e3, 3),
                "distinct_batches": n_deep, "kernels_ms": {k: round(v[0], 3) for k, v in dacc.items()},
                "k_smt": dict({"launch_ms": round(dacc["smt"][0], 3), "achieved_GBs": round(dacc["smt"][1] / (dacc["smt"][0] * 1e-3) / 1e9, 2),
                               "frac": round(dacc["smt"][1] / (dacc["smt"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}, **deep_valu(dacc["smt"][0], torch, local)),
                "state_build_s": round(t_base, 1), "batch_build_s": round(t_deep, 1),
                "note": "same step, same shape; the pre-populated state (builder.DenseState, hashed on the device) is shared by the batches, "
                        "each of which has its own L1 keys, transactions and signatures"}
    R.cs = RD = None   # (the rotation objects hold the contexts: they must go before the sweep allocates its own)
    del ctxs, c, R
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    # SURVEY 8(d) "occupancy caveat": a single 2048-transaction batch is 32 wavefronts per per-transaction kernel on a 1024-SIMD device.
    # (i) the latency of ONE batch alone, with and without HZ_FLAG_LATENCY (CU-masked chains); (ii) throughput against batches per launch,
    # two contexts in flight, up to the headline's own point; (iii) what binds at each point.
    sweep, single = None, None
    if world == 1 and not args.no_sweep:
        def small(bp, nctx, flags, steps, kernels=None):
            cs = [L.ctx("rollup-main", nTx=nTx, nLevels=lv, maxL1Tx=m1, maxFeeTx=F, device=local, n_instances=bp, flags=flags) for _ in range(nctx)]
            RS = Rotation(cs, [streams[k % len(streams)] for k in range(nctx)], bp, n_distinct, pbytes, expected, pin2, dsrc)
            RS.load()
            for k, cc in enumerate(cs):
                RS.enqueue(k)
                cc.check()
            if not args.no_verify:
                RS.verify("sweep, %d x %d" % (bp, nctx), instances=(bp - 1,))
            for k in range(nctx):
                if dsrc:
                    RS.stage(k, dsrc)
            RS.go(nctx, dsrc)
            t = D.timed(lambda: RS.go(steps, dsrc))
            RS.drain()
            if kernels is not None:   # each kernel of this shape alone on the device: which chains set the latency
                cs[0].set_profiling(True, exclusive=True)
                for _ in range(2):
                    if dsrc:
                        RS.stage(0, dsrc)
                        torch.cuda.synchronize()
                    RS.enqueue(0)
                    cs[0].check()
                    for name, ms, _by, _units in cs[0].profile():
                        kernels[name] = round(kernels.get(name, 0.0) + ms / 2, 3)
                cs[0].set_profiling(False)
            RS.cs = None
            del cs, RS
            return t / steps
        single = {}
        for key, flags in (("default", 0), ("latency_flag", 2), ("latency_solo_flags", 6)):   # 6 = HZ_FLAG_LATENCY | HZ_FLAG_SOLO
            kern_ms = {} if flags else None
            single[key] = round(small(1, 1, flags, 6, kern_ms) * 1e3, 3)
            if flags:
                single["kernels_ms_" + key] = kern_ms
        single["critical_chains"] = ("front -> eddsa (signature prologue + 148-step ladder) -> eddsa_final, and front -> hash4 -> smt (33 dependent level "
                                     "hashes) -> rtx_back -> hash_inputs: each kernel is a few dozen wavefronts, its time is the length of its dependent chain")
        sweep = []
        for bp in (1, 2, 4, 8, 16):
            if bp >= Bp or 2 * per_batch_bytes(bp) * bp > free_b - (6 << 30):
                continue
            ms = small(bp, 2, 0, 8) * 1e3
            sweep.append({"batches_per_launch": bp, "contexts": 2, "ms_per_step": round(ms, 3), "tx_per_s": round(nTx * bp / ms * 1e3, 1)})
            if bp <= 4 and 4 * per_batch_bytes(bp) * bp <= free_b - (6 << 30):
                # the latency regime's own schedule: HZ_FLAG_LATENCY contexts (every concurrent chain on its own share of the compute
                # units), two of them in flight -- plain contexts do not overlap there (their kernels evict each other's code from the
                # instruction caches of the CUs they share: profiles/r05_latency_regime.txt).
                # Each point in a process of its own: many such queues in a process that has made other contexts before can abort in
                # the runtime (csrc/ctx.hip "the other half of the same hazard"), and the main line must not depend on that.
                # (four in flight reach 560 k / 859 k / 1 118 k tx/s at 1 / 2 / 4 batches in a process that has the device to itself --
                # tools/experiments/latency_inflight.sh, profiles/r05_latency_regime.txt -- but sixteen CU-masked queues beside this
                # process's own exceed the hardware queues of the device and the scheduler then time-slices them: not run here)
                # (with an outer process -- the default run, with_node_host -- both points are measured by IT after this worker has exited:
                #  a process that has the device to itself, as a coordinator's would; a child of this process runs beside this process's
                #  idle queues and loses ~7 %: 413 k against 448 k tx/s at 1 x 2)
                if args.gpu_worker:
                    sweep[-1]["latency_flag_x2"] = None
                else:
                    sweep[-1]["latency_flag_x2"] = flagged_point(bp, 2)
                # (four in flight -- the library's cap on partitioned contexts since round 6 -- are sixteen CU-masked queues: beside THIS
                #  process's own queues the scheduler time-slices them (25 k tx/s instead of 560 k), so those points are measured by the
                #  outer process after this worker has exited: with_node_host)
        sweep.append({"batches_per_launch": Bp, "contexts": inflight, "ms_per_step": round(dt / args.steps * 1e3, 3), "tx_per_s": round(nTx * Bp * args.steps / dt, 1)})
        for i, pt in enumerate(sweep):
            # doubling the batches of a launch: a step that barely gets longer is waiting on dependent chains (latency); one that
            # doubles is out of issue slots (the integer pipe: see roofline_valu) -- HBM never binds this pass (whole_pass.frac)
            if i + 1 < len(sweep) and sweep[i + 1]["batches_per_launch"] == 2 * pt["batches_per_launch"]:
                g = sweep[i + 1]["ms_per_step"] / pt["ms_per_step"]
                pt["step_growth_when_doubled"] = round(g, 3)
                pt["binds"] = "latency" if g < 1.35 else ("latency -> valu" if g < 1.75 else "valu")
            else:
                pt["binds"] = "valu"
    L.host_free(pin)
    del dsrc_t
    L.host_free(pin2)
    torch.cuda.empty_cache()

    out = None
    if rank == 0:
        total_tx = nTx * Bp * args.steps * world
        value = total_tx / dt
        # dominant kernel = most GPU time per step over all its launches (k_smt runs for the transactions and for the
        # fee transactions); its roofline is quoted on the transaction launch
        kern = {"smt": "k_smt", "fee_smt": "k_smt", "hash4": "k_hash4", "fee_hash": "k_hash4", "eddsa": "k_eddsa", "eddsa_fix": "k_eddsa_fix", "front": "k_main_front"}
        tot, byt = {}, {}
        for name, v in acc.items():
            tot[kern.get(name, name)] = tot.get(kern.get(name, name), 0.0) + v[0]
            byt[kern.get(name, name)] = byt.get(kern.get(name, name), 0) + v[1]
        # dominant kernel for an HBM roofline = the kernel that writes most of the witness (k_smt: three quarters of a step's
        # bytes, and the only one besides the front / hash kernels that fills the device). k_eddsa's launches last longer
        # but run on few wavefronts, latency bound, underneath the others: their figures are in kernels_ms / kernels_GBs.
        dk = max(tot, key=lambda k: byt[k])
        dname = max((n for n in acc if kern.get(n, n) == dk), key=lambda n: acc[n][0])
        dms, dbytes, dunits, dlaunches = acc[dname]
        dlaunches = max(1, int(round(dlaunches)))
        achieved = dbytes / (dms * 1e-3) / 1e9
        # which roofline binds, from measured fractions: the HBM side is this run's bytes / time; the integer side prices the VALU
        # wave-instructions the committed SQ_INSTS_VALU pass counted for this command line at the issue cost of the instruction mix
        props = torch.cuda.get_device_properties(local)
        n_simd = props.multi_processor_count * 4
        clock_hz = float(getattr(props, "clock_rate", 0) or 2400000) * 1e3
        frac_hbm = achieved / HBM_PEAK_GBS
        k_insts, vmeta = measured_valu(dk)
        frac_valu = k_insts * VALU_CYCLES_PER_INST / (n_simd * clock_hz * (dms / dlaunches) * 1e-3) if k_insts else None
        step_insts, _ = measured_valu(None)
        roofline_valu = None
        if step_insts:
            step_s = dt / args.steps
            roofline_valu = {"insts_per_step": int(step_insts), "cycles_per_inst": VALU_CYCLES_PER_INST, "simds": n_simd, "clock_GHz": round(clock_hz / 1e9, 3),
                             "frac": round(step_insts * VALU_CYCLES_PER_INST / (n_simd * clock_hz * step_s), 5),
                             "measured_cycles_per_inst": round(n_simd * clock_hz * step_s / step_insts, 3),
                             "kernel": dk, "kernel_insts_per_launch": int(k_insts) if k_insts else None, "kernel_frac": round(frac_valu, 5) if frac_valu else None,
                             "source": "SQ_INSTS_VALU of %s (%s), x %.1f cycles per wave-instruction of this mix (profiles/r03_instbench.txt), "
                                       "/ (SIMDs x clock x the step time of THIS run)" % (vmeta.get("source_file", "?") if vmeta else "?", vmeta.get("command", "?") if vmeta else "?", VALU_CYCLES_PER_INST)}
        bound = "valu" if (frac_valu or 0) > frac_hbm else "hbm"
        out = {
            "metric": "rollup-main tx-witnesses/sec (nTx=%d, nLevels=%d)" % (nTx, lv),
            "value": round(value, 1), "unit": "tx-witnesses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "rollup-main nTx=%d nLevels=%d maxL1Tx=%d maxFeeTx=%d" % (nTx, lv, m1, F), "batches_per_launch": Bp, "contexts_in_flight": inflight,
                       "distinct_batches": n_distinct,
                       "rotation": ("none (--no-rotate): every step re-evaluates the inputs its instances hold" if args.no_rotate else
                                    "every step a new round: instance b of context k holds batch (k * %d + b + round) mod %d; the round's packed inputs (resident in HBM) are "
                                    "scattered into the witness layout inside the timed region; hashGlobalInputs of the last round checked after it" % (Bp, n_distinct)),
                       "constant_marks": os.environ.get("HZ_NO_ZMARK") is None, "state_accounts": n_acc, "l1_txs": nTx - n_l2, "l2_signed_txs": n_l2, "parallelism": "batch-dp%d" % world,
                       "world_size": world, "backend": D.backend if world > 1 else None,
                       "witness_bytes_per_batch": witness_bytes, "step_latency_ms": round(single_ms, 3), "batch_build_s": round(t_build, 1),
                       "batch_builder": ({"kind": "native (libhz_host.so hzb_batch_build_begin / _finish) + hz_poseidon_dag", "batches": len(all_seeds),
                                          "pipelined": "the device evaluates batch i's Merkle hashes (a worker thread) while the host walks batch i + 1",
                                          "hashes": builder_stats["jobs"], "dag_segments": builder_stats["segments"], "device_ms": round(builder_stats["device_ms"], 1),
                                          "walk_and_sign_s": round(builder_stats["walk_s"], 2), "evaluator_s": round(builder_stats["eval_s"], 2),
                                          "state_s": round(builder_stats["state_s"], 2), "batch_s": round(builder_stats["batch_s"], 2),
                                          "phases_ms_per_batch": {k: round(1e3 * v / max(1, len(all_seeds)), 2) for k, v in builder_stats.get("phases_s", {}).items()},
                                          "recipe": "native (hzb_batch_add_synthetic: the reference generator's recipe inside libhz_host.so)",
                                          "ms_per_batch": round(1e3 * builder_stats["batch_s"] / max(1, len(all_seeds)), 1),
                                          "sign_ms_per_batch": round(1e3 * builder_stats["sign_s"] / max(1, len(all_seeds)), 1),
                                          "ms_per_batch_without_signing": round(1e3 * (builder_stats["batch_s"] - builder_stats["sign_s"]) / max(1, len(all_seeds)), 1),
                                          "builder_tx_per_s": round(nTx * len(all_seeds) / builder_stats["batch_s"], 1),
                                          "builder_tx_per_s_without_signing": round(nTx * len(all_seeds) / max(1e-9, builder_stats["batch_s"] - builder_stats["sign_s"]), 1),
                                          "pipeline_ratio": round(nTx * len(all_seeds) / builder_stats["batch_s"] / value, 4),
                                          "note": "one host process; from the seed to the packed inputs in pinned memory (transactions, walk, signing, Merkle hashing as one "
                                                  "DAG on the GPU, packing); signing is a wallet's work in production; pipeline_ratio = builder_tx_per_s / value"}
                                         if builder_stats else {"kind": "python (circuits_amd/builder.py), host hashing, process pool", "batches": len(all_seeds)})},
            "roofline": {"bound": bound, "kernel": dk, "launch": dname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_hbm": round(frac_hbm, 5), "frac_valu": round(frac_valu, 5) if frac_valu else None,
                         "traffic": measured_traffic(dk, Bp)[0],
                         "stall": stall_attribution(dk),
                         "traffic_source": "%s (committed PMC pass of this command line, not measured by this run)" % measured_traffic(dk, Bp)[1] if measured_traffic(dk, Bp)[1] else None,
                         "launches_per_step": dlaunches, "launch_ms": round(dms / dlaunches, 3),
                         "algorithmic_bytes_per_launch": int(dbytes // dlaunches), "gpu_ms_per_step_all_launches": round(tot[dk], 3),
                         "bytes_per_launch_into_a_fresh_buffer": int(full_bytes.get(dname, 0)),
                         "left_in_place_frac": round(1.0 - dbytes / dlaunches / full_bytes[dname], 4) if full_bytes.get(dname) else None,
                         "note": "achieved / peak / frac price the kernel against the HBM roofline (the contract's figure); `bound` names the roofline with "
                                 "the larger measured fraction: frac_valu = its VALU wave-instructions x issue cycles / (SIMDs x clock x its duration). "
                                 "algorithmic bytes = 32 B x the witness signals this launch is responsible for; duration = HIP events on its stream with the "
                                 "kernel alone on the device; traffic = FETCH_SIZE + WRITE_SIZE of the committed PMC passes. "
                                 "Constant marks (DESIGN.md 4): the witness buffer is persistent, and the levels of an SMT proof above its leaf get the same "
                                 "constant S-box block whatever the batch -- k_smt keeps, per chain and unit, the level from which the buffer already holds it "
                                 "and stores an empty level only below that mark. algorithmic_bytes_per_launch counts what the launch DID store (the kernel "
                                 "counts what it left in place: left_in_place_frac of bytes_per_launch_into_a_fresh_buffer), measured on steps that meet new "
                                 "batches (config.rotation). What is left is integer-VALU issue bound: frac_valu is the binding fraction"},
            "whole_pass": {"algorithmic_bytes_per_tx": algorithmic_bytes_per_tx(lv, F),
                           "achieved_GBs": round(algorithmic_bytes_per_tx(lv, F) * value / world / 1e9, 2),
                           "frac": round(algorithmic_bytes_per_tx(lv, F) * value / world / 1e9 / HBM_PEAK_GBS, 5)},
            "kernels_ms": {k: round(v[0], 3) for k, v in acc.items()},
            "kernels_GBs": {k: round(v[1] / (v[0] * 1e-3) / 1e9, 1) for k, v in acc.items() if v[0] > 0},
        }
        if roofline_valu is not None:
            out["roofline_valu"] = roofline_valu
        if single is not None:
            out["single_batch_latency_ms"] = dict(single, note="one 2048-transaction batch alone on the device, enqueue + check, mean of 6; latency_flag = HZ_FLAG_LATENCY "
                                                              "(the context's concurrent chains on disjoint compute units); latency_solo_flags = with HZ_FLAG_SOLO as well (the SMT chain "
                                                              "kernel in its latency form: right only when no other context is in flight)")
            out["batches_sweep"] = sweep
        if standin is not None:
            out["shard_tx_standin"] = standin
        if export is not None:
            for k in ("export_ms_per_batch", "export_ms_per_batch_4_per_call", "value_export", "value_export_4_per_call", "value_delivered_host", "wtns_write_s"):
                if k in export:
                    out[k] = export[k]
            out["export"] = export
        if deep is not None:
            deep["ratio_to_value"] = round(deep["value"] / value, 4)
            # what a hashing level costs: the two committed SQ_INSTS_VALU passes differ by the levels between the two states' leaves
            try:
                i_deep, i_base = float(deep["k_smt"]["insts_valu_per_launch"]), float(k_insts)
                waves = nTx * Bp * 4 / 64.0                                   # wavefronts of the transaction launch (four chains per transaction)
                lv_base, lv_deep = (n_acc - 1).bit_length(), deep["state_accounts"].bit_length() - 1
                per_level = (i_deep - i_base) / (waves * (lv_deep - lv_base))
                deep["k_smt"].update({"insts_per_hashing_level_and_wave": int(per_level), "hashing_levels": [lv_base, lv_deep],
                                      "insts_per_wave": [int(i_base / waves), int(i_deep / waves)],
                                      "not_level_hashes_frac": [round(1 - per_level * (lv_base + 0.9) * waves / i_base, 4), round(1 - per_level * (lv_deep + 0.9) * waves / i_deep, 4)],
                                      "note": "per chain-wavefront: (instructions on the deep state - on the recipe's state) / (levels between their leaves); a proof "
                                              "reaches its leaf about 0.9 levels below log2(accounts); not_level_hashes_frac = the share of a launch's instructions that is "
                                              "NOT level hashing (key bits, SMTLevIns inversions, state machine, conversions, stores of the dead levels)"})
            except (KeyError, TypeError, ZeroDivisionError):
                pass
            out["value_deep_state"] = deep["value"]
            out["deep_state"] = deep
        if dt_e2e is not None:
            out["value_e2e"] = round(total_tx / dt_e2e, 1)
            out["e2e"] = {"ms_per_step": round(dt_e2e / args.steps * 1e3, 3), "ratio_to_value": round(dt / dt_e2e, 4), "packed_input_bytes_per_batch": pbytes,
                          "upload_ms_per_batch": round(upload_ms, 4), "upload_GBs": round(pbytes / upload_ms / 1e6, 2) if upload_ms else None,
                          "note": "timed region = per step and batch one hz_inputs_stage from pinned host memory (async H2D beside the previous step's kernels) "
                                  "+ unpack kernel + witness kernels + check; upload_* = hz_inputs_upload (copy + unpack) alone on the device"}
    # secondary lines: config 4 sharded (N > 1), config 5 withdraw, Poseidon-BN254/sec, CPU baseline
    if world > 1 and not args.no_shard:
        # The secondary line must never cost the main one: RCCL over xGMI has not run on any box this code was built on. A failure is
        # recorded in the line; a collective that hangs is cut off by a watchdog that prints the main line and ends the rank.
        import threading

        def _give_up():
            if rank == 0:
                out["shard_tx"] = {"error": "tx-sharded pass did not finish within %d s" % args.shard_timeout}
                print(json.dumps(out), flush=True)
            os._exit(0)
        wd = threading.Timer(args.shard_timeout + (0 if rank == 0 else 15), _give_up)
        wd.daemon = True
        wd.start()
        try:
            sh = bench_sharded(args, L, D, shared[0], shared[1])
        except Exception as e:   # noqa: BLE001 -- reported, not raised
            sh = {"error": "%s: %s" % (type(e).__name__, e)}
        wd.cancel()
        if rank == 0:
            out["shard_tx"] = sh
    if world == 1 and not args.no_withdraw:
        out["withdraw"] = bench_withdraw(args, L, D, launches=max(1, args.withdraw_total // args.withdraw_per_launch), steps=2)
    if rank == 0:
        if world == 1 and not args.no_poseidon:
            out["poseidon_bn254"] = poseidon_rates(L, torch)
        if world == 1 and args.cpu_sample > 0:
            n_cpu = min(args.cpu_sample, nTx)
            workers = args.cpu_workers
            if workers <= 0:   # every CPU this process may use, as far as a third of its memory goes (oracle witness + builder per process)
                per_proc = 32 * 60000 * n_cpu * 1.15 + (1 << 30)
                cpus, mem = host_limits()
                workers = max(1, min(cpus, int(mem // 3 // per_proc)))
            out["cpu_baseline"] = cpu_baseline(n_cpu, lv, min(m1, max(1, n_cpu // 8)), F, workers)
        print(json.dumps(out))
    D.close()


if __name__ == "__main__":
    main()
