"""Intra-batch sharding of one RollupMain batch over the GPUs of a node (BASELINE config 4).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI). Rank r evaluates the
transactions hz_shard_range(nTx, world, r); the only exchange is one all_gather of the per-
transaction data-availability records (160 B each, ~328 KB for 2048 txs -- latency bound, the xGMI
link bandwidth is irrelevant), after which rank 0 evaluates the fee transactions and HashInputs.
The witness stays sharded in HBM: rank r holds the signals of its transactions, rank 0 in addition
the fee-tx and HashInputs sections.
"""


def shard_ranges(lib, n_tx, world):
    return [lib.shard_range(n_tx, world, r) for r in range(world)]


def max_shard(ranges):
    return max(c for _, c in ranges)


class ShardedBatch:
    """Drives one sharded witness pass. `ctx` is a RollupMain Ctx (or any object with the same
    set_shard/enqueue/da_export/da_import/enqueue_tail/check methods); `alloc(nbytes)` returns a
    device byte buffer object with .data_ptr(); `all_gather(out_list_buf, in_buf)` gathers equal
    sized buffers from all ranks into rank order."""

    def __init__(self, ctx, lib, n_tx, rank, world, alloc, all_gather):
        self.ctx, self.rank, self.world, self.all_gather = ctx, rank, world, all_gather
        self.ranges = shard_ranges(lib, n_tx, world)
        self.first, self.count = self.ranges[rank]
        self.rec = ctx.da_record_bytes()
        self.slot = max_shard(self.ranges) * self.rec
        self.send = alloc(self.slot)
        self.recv = alloc(self.slot * world)
        ctx.set_shard(self.first, self.count, rank == 0)

    def step(self, stream):
        """One sharded pass on HIP stream `stream` (a hipStream_t handle, required): export, collective, imports and the tail
        are ordered by that stream, so `all_gather` must enqueue the collective on it too (bench.py: `with torch.cuda.stream(s)`),
        or block until `send` is complete and `recv` filled."""
        if not stream:
            raise ValueError("ShardedBatch.step needs an explicit stream: the collective has to be ordered with the export/import kernels")
        c = self.ctx
        c.enqueue(stream)                       # this rank's transactions
        c.da_export(self.send.data_ptr(), stream)
        self.all_gather(self.recv, self.send)   # the single collective of the path
        if self.rank == 0:
            for r in range(1, self.world):
                f, n = self.ranges[r]
                c.da_import(f, n, self.recv.data_ptr() + r * self.slot, stream)
            c.enqueue_tail(stream)              # FeeTx + HashInputs (SHA-256) on rank 0
        c.check()
