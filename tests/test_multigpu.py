"""Multi-GPU path: (CPU) world_size-2 gloo run of the shard orchestration with a recording fake
context; (GPU) the same orchestration on one device with two contexts standing in for two ranks,
checked bit-exact against the oracle."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_partition_the_batch():
    from circuits_amd import lib
    L = lib()
    for n_tx, world in ((2048, 8), (2048, 1), (10, 4), (3, 8), (256, 2)):
        rs = [L.shard_range(n_tx, world, r) for r in range(world)]
        pos = 0
        for f, c in rs:
            assert f == pos and c >= 0
            pos += c
        assert pos == n_tx
        assert max(c for _, c in rs) - min(c for _, c in rs) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tx, out):
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from circuits_amd import lib
    from circuits_amd.multigpu import ShardedBatch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Buf:
        def __init__(self, n):
            self.t = torch.zeros(n, dtype=torch.uint8)

        def data_ptr(self):
            return self.t.data_ptr()

    class FakeCtx:
        """records the calls and moves recognisable bytes through the exchange"""
        def __init__(self):
            self.log = []
            self.imported = {}

        def da_record_bytes(self):
            return 160

        def set_shard(self, f, n, tail):
            self.first, self.count, self.tail = f, n, tail

        def enqueue(self, stream=None):
            self.log.append("enqueue")

        def da_export(self, ptr, stream=None):
            self.log.append("export")
            for k in range(self.count):
                self.send.t[k * 160:(k + 1) * 160] = (self.first + k) % 251

        def da_import(self, f, n, ptr, stream=None):
            self.log.append(("import", f, n))
            off = ptr - self.recv.data_ptr()
            for k in range(n):
                rec = self.recv.t[off + k * 160: off + (k + 1) * 160]
                assert int(rec[0]) == (f + k) % 251 and int(rec[159]) == (f + k) % 251
                self.imported[f + k] = True

        def enqueue_tail(self, stream=None):
            self.log.append("tail")

        def enqueue_tail_chain(self, stream=None):
            self.log.append("tail_chain")

        def sha_blocks(self):
            return 11

        def sha_state_bytes(self):
            return 11 * 64 + 12 * 32

        def sha_export(self, ptr, stream=None):
            self.log.append("sha_export")
            self.sha.t[:] = 7   # recognisable state

        def sha_expand(self, first, count, ptr, stream=None):
            assert (ptr is None) == (rank == 0)
            assert int(self.sha.t[0]) == 7 and int(self.sha.t[-1]) == 7   # the broadcast arrived before the expansion
            self.log.append(("sha_expand", first, count))

        def check(self):
            self.log.append("check")

    ctx = FakeCtx()
    bufs = []

    def alloc(n):
        b = Buf(n)
        bufs.append(b)
        return b

    def all_gather(recv, send):
        parts = [torch.zeros_like(send.t) for _ in range(world)]
        dist.all_gather(parts, send.t)
        recv.t.copy_(torch.cat(parts))

    def broadcast(buf):
        dist.broadcast(buf.t, src=0)

    sb = ShardedBatch(ctx, lib(), n_tx, rank, world, alloc, all_gather, broadcast)
    ctx.send, ctx.recv, ctx.sha = sb.send, sb.recv, sb.sha
    sb.step(1)   # the fake context ignores the stream handle
    ok = ctx.log[0] == "enqueue" and ctx.log[1] == "export" and ctx.log[-1] == "check"
    # the SHA-256 blocks are split over the ranks: 11 blocks -> 6 + 5
    ok = ok and ctx.log[-2] == ("sha_expand", 0, 6) if rank == 0 else ok and ctx.log[-2] == ("sha_expand", 6, 5)
    if rank == 0:
        ok = ok and ctx.log[-4:-2] == ["tail_chain", "sha_export"] and "tail" not in ctx.log and len(ctx.imported) == n_tx - sb.count and ctx.tail
    else:
        ok = ok and "tail" not in ctx.log and "tail_chain" not in ctx.log and not ctx.tail
    res = torch.tensor([1 if ok else 0])
    dist.all_reduce(res, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out, "w").write(str(int(res.item())))
    dist.destroy_process_group()


def test_sharded_orchestration_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "ok")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, 10, out), nprocs=2, join=True)
    assert open(out).read() == "1"


@pytest.mark.gpu
def test_sharded_batch_on_one_gpu_matches_oracle(hz):
    """Two contexts stand in for two ranks on one device; the union of their witnesses and rank 0's
    public output must equal the unsharded oracle witness."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleCtx
    from circuits_amd import builder as B
    from circuits_amd.multigpu import ShardedBatch
    n_tx = 8
    bb = B.synthetic_batch(n_tx, 16, 3, 4, n_accounts=6, exits=2)
    inp = bb.get_input()
    world = 2
    ctxs = [hz.ctx("rollup-main", nTx=n_tx, nLevels=16, maxL1Tx=3, maxFeeTx=4) for _ in range(world)]
    for c in ctxs:
        c.set_inputs(inp)
    mailbox = {}

    def alloc(n):
        t = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # filled before any other stream touches it
        return t

    sbs = []
    streams = [torch.cuda.Stream() for _ in range(world)]
    for r in range(world):
        def all_gather(recv, send, r=r):
            with torch.cuda.stream(streams[r]):   # ordered with the export / import kernels of this "rank"
                mailbox[r] = send.clone()
                if r == 0:  # rank 0 runs last in this single-process stand-in: every send is in the mailbox
                    streams[0].wait_stream(streams[1])
                    recv.copy_(torch.cat([mailbox[k] for k in range(world)]))
        sbs.append(ShardedBatch(ctxs[r], hz, n_tx, r, world, alloc, all_gather))
    for r in (1, 0):
        sbs[r].step(streams[r].cuda_stream)
    assert ctxs[0].get("main.hashGlobalInputs") == bb.get_hash_inputs()
    o = OracleCtx("rollup-main", n_tx, 16, 3, 4)
    o.set_inputs(inp)
    assert o.run() is None
    ob = o.read_raw_bytes()
    gb = [c.read_raw_bytes() for c in ctxs]
    # per-transaction section: unit u's signals must come from the rank that owns u
    # (physical layout is signal-major: element index = base + sig * nTx + unit)
    idx0 = ctxs[0].lookup("main.decodeTx[0].n2bData.out[0]")
    idx_last = ctxs[0].lookup("main.rollupTx[0].s5.out")
    for r in range(world):
        f, n = sbs[r].first, sbs[r].count
        for e in range(idx0, idx_last + n_tx, n_tx):
            for u in range(f, f + n):
                a = 32 * (e + u)
                assert gb[r][a:a + 32] == ob[a:a + 32], (r, e, u)
    # fee-tx and hash-inputs sections on rank 0
    fee0 = ctxs[0].lookup("main.feeTx[0].feeIdxIsZero.inv")
    assert gb[0][32 * fee0:] == ob[32 * fee0:]


@pytest.mark.gpu
def test_sharded_batch_split_sha_tail_matches_oracle(hz):
    """The tail split over the ranks (SURVEY 8e "scatter blocks back"): rank 0 runs FeeTx, the message and the sequential SHA-256
    chain, its (message, chaining values) reach the other rank through the broadcast, and each rank writes the bit-level witness of
    its own range of blocks. Two contexts stand in for two ranks; rank 0's public output is the builder's, and every byte of the
    HashInputs section matches the oracle on the rank that owns it -- the blocks a rank does not own it must not have written."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleCtx
    from circuits_amd import builder as B
    from circuits_amd.multigpu import ShardedBatch
    shape = (24, 16, 6, 4)
    n_tx = shape[0]
    bb = B.synthetic_batch(*shape, n_accounts=10, exits=2, seed=31)
    inp = bb.get_input()
    world = 2
    ctxs = [hz.ctx("rollup-main", nTx=n_tx, nLevels=16, maxL1Tx=6, maxFeeTx=4) for _ in range(world)]
    for c in ctxs:
        c.set_inputs(inp)
    streams = [torch.cuda.Stream() for _ in range(world)]
    mailbox, shabox = {}, {}

    def alloc(n):
        t = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # filled before any other stream touches it
        return t

    sbs = []
    for r in range(world):
        def all_gather(recv, send, r=r):
            with torch.cuda.stream(streams[r]):
                mailbox[r] = send.clone()
                if r == 0:
                    streams[0].wait_stream(streams[1])
                    recv.copy_(torch.cat([mailbox[k] for k in range(world)]))

        def broadcast(buf, r=r):
            with torch.cuda.stream(streams[r]):
                if r == 0:
                    shabox[0] = buf.clone()
                else:
                    streams[r].wait_stream(streams[0])
                    buf.copy_(shabox[0])
        sbs.append(ShardedBatch(ctxs[r], hz, n_tx, r, world, alloc, all_gather, broadcast))
    # single-process stand-in: rank 1's first half (its transactions, its export), then rank 0's whole step, then rank 1's expansion
    c1, s1 = ctxs[1], streams[1].cuda_stream
    c1.enqueue(s1)
    c1.da_export(sbs[1].send.data_ptr(), s1)
    sbs[1].all_gather(sbs[1].recv, sbs[1].send)
    sbs[0].step(streams[0].cuda_stream)
    sbs[1].broadcast(sbs[1].sha)
    c1.sha_expand(sbs[1].blocks[0], sbs[1].blocks[1], sbs[1].sha.data_ptr(), s1)
    c1.check()
    assert ctxs[0].get("main.hashGlobalInputs") == bb.get_hash_inputs()
    o = OracleCtx("rollup-main", *shape)
    o.set_inputs(inp)
    assert o.run() is None
    ob = o.read_raw_bytes()
    gb = [c.read_raw_bytes() for c in ctxs]
    nb = ctxs[0].sha_blocks()
    assert sbs[0].blocks[1] + sbs[1].blocks[1] == nb and sbs[1].blocks[0] == sbs[0].blocks[1] and nb >= 8
    # block b's witness = the per_block signals named sha256compression[b].* (contiguous in the symbol table, block after block)
    names = [n for n in o.symbol_names() if ".inputsHasher.sha256compression[" in n]
    first = o.lookup("main.hasherInputs.inputsHasher.sha256compression[0].sigmaPlus[0].sigma0.xor3.mid[0]")
    assert len(names) % nb == 0
    per_block = len(names) // nb
    for blk in range(nb):
        owner = 0 if blk < sbs[0].blocks[1] else 1
        lo_i, hi_i = first + blk * per_block, first + (blk + 1) * per_block
        assert gb[owner][32 * lo_i:32 * hi_i] == ob[32 * lo_i:32 * hi_i], (blk, owner)
        assert gb[1 - owner][32 * lo_i:32 * hi_i] != ob[32 * lo_i:32 * hi_i], "block %d also written by rank %d" % (blk, 1 - owner)


@pytest.mark.gpu
def test_config4_eight_shards_on_one_gpu_match_oracle(hz, config4):
    """BASELINE config 4 at its literal shape, partitioned as the north star says -- RollupMain(2048, 32, 256, 64), 8 ranks x 256
    transactions, the 766 SHA-256 blocks split 8 ways (reference src/rollup-main.circom:93-99,433-474) -- with eight contexts standing
    in for the eight ranks on one device (no 8-GPU node is available to the builds; the exchange buffers are the real hz_da_export /
    hz_sha_export records). The UNION of the shards is the oracle's witness: every transaction's signals on the rank that owns it,
    every SHA-256 block written by exactly its owner, the fee transactions and the public output on rank 0."""
    import numpy as np
    import torch
    from circuits_amd.multigpu import ShardedBatch
    shape, bb, inp, o = config4["shape"], config4["batch"], config4["input"], config4["oracle"]
    n_tx, world = shape[0], 8
    ctxs = [hz.ctx("rollup-main", nTx=n_tx, nLevels=shape[1], maxL1Tx=shape[2], maxFeeTx=shape[3]) for _ in range(world)]
    for name in inp:   # one host pass per signal, then device-to-device onto the other "ranks"
        ctxs[0].set_input(name, inp[name])
    lay = ctxs[0].packed_layout()
    from circuits_amd.capi import pack_inputs
    packed = pack_inputs(lay, inp)
    for c in ctxs[1:]:
        c.upload(0, packed)
    streams = [torch.cuda.Stream() for _ in range(world)]
    mailbox, shabox = {}, {}

    def alloc(n):
        t = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()   # filled before any other stream touches it
        return t
    sbs = []
    for r in range(world):
        def all_gather(recv, send, r=r):
            with torch.cuda.stream(streams[r]):
                mailbox[r] = send.clone()
                if r == 0:   # rank 0 runs after the others' first halves in this single-process stand-in
                    for k in range(1, world):
                        streams[0].wait_stream(streams[k])
                    recv.copy_(torch.cat([mailbox[k] for k in range(world)]))

        def broadcast(buf, r=r):
            with torch.cuda.stream(streams[r]):
                if r == 0:
                    shabox[0] = buf.clone()
                else:
                    streams[r].wait_stream(streams[0])
                    buf.copy_(shabox[0])
        sbs.append(ShardedBatch(ctxs[r], hz, n_tx, r, world, alloc, all_gather, broadcast))
    assert [sb.count for sb in sbs] == [256] * 8 and sbs[0].rec * 2048 == sbs[0].slot * world
    for r in range(1, world):   # the other ranks' first halves: their transactions, their export
        c, s = ctxs[r], streams[r].cuda_stream
        c.enqueue(s)
        c.da_export(sbs[r].send.data_ptr(), s)
        sbs[r].all_gather(sbs[r].recv, sbs[r].send)
    sbs[0].step(streams[0].cuda_stream)
    for r in range(1, world):   # their second halves: the broadcast, their share of the blocks
        sbs[r].broadcast(sbs[r].sha)
        ctxs[r].sha_expand(sbs[r].blocks[0], sbs[r].blocks[1], sbs[r].sha.data_ptr(), streams[r].cuda_stream)
        ctxs[r].check()
    assert ctxs[0].get("main.hashGlobalInputs") == bb.get_hash_inputs()
    nb = ctxs[0].sha_blocks()
    assert nb == 766 and sum(sb.blocks[1] for sb in sbs) == nb and all(sbs[r].blocks[0] == sum(sb.blocks[1] for sb in sbs[:r]) for r in range(world))
    # transactions: the computed signals of the tx section (signal-major: element = first + sig * nTx + unit), unit u on the rank that owns u
    idx0 = ctxs[0].lookup("main.decodeTx[0].n2bData.out[0]")
    idx_end = ctxs[0].lookup("main.rollupTx[0].s5.out") + n_tx
    rows = (idx_end - idx0) // n_tx
    assert (idx_end - idx0) % n_tx == 0 and rows > 40000
    for r0 in range(0, rows, 2048):
        nr = min(2048, rows - r0)
        ref = np.frombuffer(o.read_raw_bytes(idx0 + r0 * n_tx, nr * n_tx), dtype=np.uint8).reshape(nr, n_tx, 32)
        for r in range(world):
            f, n = sbs[r].first, sbs[r].count
            got = np.frombuffer(ctxs[r].read_raw_bytes(idx0 + r0 * n_tx, nr * n_tx), dtype=np.uint8).reshape(nr, n_tx, 32)
            assert np.array_equal(got[:, f:f + n, :], ref[:, f:f + n, :]), "rank %d, signal rows from %d" % (r, r0)
    # SHA-256 blocks: exactly the owner holds a block's witness
    first = o.lookup("main.hasherInputs.inputsHasher.sha256compression[0].sigmaPlus[0].sigma0.xor3.mid[0]")
    names = [n for n in o.symbol_names() if ".inputsHasher.sha256compression[" in n]
    per_block = len(names) // nb
    assert len(names) % nb == 0
    for r in range(world):
        b0, bn = sbs[r].blocks
        for owner_side in (True, False):
            if owner_side:
                lo_i, cnt = first + b0 * per_block, bn * per_block
                assert ctxs[r].read_raw_bytes(lo_i, cnt) == o.read_raw_bytes(lo_i, cnt), "rank %d: its own blocks" % r
            else:
                for blk in ((b0 + bn) % nb, (b0 - 1) % nb):   # a neighbour's block on either side: not written here
                    lo_i = first + blk * per_block
                    assert ctxs[r].read_raw_bytes(lo_i, per_block) != o.read_raw_bytes(lo_i, per_block), "block %d also written by rank %d" % (blk, r)
    # fee transactions and the rest of HashInputs (bit packing, the digest) on rank 0
    fee0 = ctxs[0].lookup("main.feeTx[0].feeIdxIsZero.inv")
    assert ctxs[0].read_raw_bytes(fee0, first - fee0) == o.read_raw_bytes(fee0, first - fee0)
    tail0 = first + nb * per_block
    assert ctxs[0].read_raw_bytes(tail0) == o.read_raw_bytes(tail0)


@pytest.mark.gpu
def test_bench_gpus8_eight_ranks_on_one_gpu(hz):
    """`python bench.py --gpus 8` as the driver launches it on an 8-GPU node, here with the one-GPU hook (all eight ranks on device 0,
    gloo collectives): eight real processes, real hz_da_export records through the all_gather and the SHA-256 state through the
    broadcast, n_gpus = 8 in the line and the tx-sharded secondary line at 5 transactions per rank."""
    import json
    import subprocess
    env = dict(os.environ, HZ_BENCH_DEVICE="0", HZ_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--nTx", "40", "--nLevels", "16", "--maxL1Tx", "8", "--maxFeeTx", "4",
           "--batches-per-launch", "2", "--inflight", "1", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--build-workers", "1", "--no-e2e"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["world_size"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    sh = line["shard_tx"]
    assert "error" not in sh, "%s\n%s" % (sh, r.stderr[-3000:])
    assert sh["scaling"] == "strong" and sh["transactions_per_rank"] == 5 and "all_gather" in sh["collective"] and "broadcast" in sh["collective"]


@pytest.mark.gpu
def test_bench_sharded_pass_through_rccl_on_one_gpu(hz):
    """the RCCL calls of the sharded pass -- all_gather_into_tensor and broadcast enqueued on the pass's own
    stream between the export / import / expansion kernels -- executed for real: a process group of backend nccl with world size 1
    (HZ_BENCH_FORCE_COLLECTIVE=1). The pass must reproduce the builder's hashGlobalInputs (bench.py asserts it)."""
    import json
    import subprocess
    env = dict(os.environ, HZ_BENCH_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HZ_BENCH_BACKEND", "HZ_BENCH_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--shard-tx", "--nTx", "40", "--nLevels", "16", "--maxL1Tx", "8", "--maxFeeTx", "4",
           "--steps", "4", "--warmup", "1", "--cpu-sample", "0", "--build-workers", "1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["value"] > 0
    assert "nccl" in line["config"]["collective"] and "broadcast" in line["config"]["collective"]


@pytest.mark.gpu
def test_bench_gpus2_spawns_two_ranks_and_exchanges_real_records(hz):
    """`python bench.py --gpus 2` launches its two ranks itself and, with the one-GPU test hook (both ranks on
    device 0, gloo collectives), moves REAL hz_da_export records between two processes: the sharded pass must reproduce the
    builder's hashGlobalInputs on rank 0 (bench.py asserts it) and the line must say n_gpus = 2."""
    import json
    import subprocess
    env = dict(os.environ, HZ_BENCH_DEVICE="0", HZ_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--nTx", "40", "--nLevels", "16", "--maxL1Tx", "8", "--maxFeeTx", "4",
           "--batches-per-launch", "2", "--inflight", "1", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--build-workers", "1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["world_size"] == 2 and line["scaling"] == "weak"
    assert line["config"]["distinct_batches"] == 2 and line["value"] > 0 and line["value_e2e"] > 0
    sh = line["shard_tx"]
    assert sh["scaling"] == "strong" and sh["transactions_per_rank"] == 20 and "all_gather" in sh["collective"]
