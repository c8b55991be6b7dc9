"""Intra-batch sharding of one RollupMain batch over the GPUs of a node (BASELINE config 4).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI). Rank r evaluates the
transactions hz_shard_range(nTx, world, r); the exchange is one all_gather of the per-
transaction data-availability records (160 B each, ~328 KB for 2048 txs -- latency bound, the xGMI
link bandwidth is irrelevant), after which rank 0 evaluates the fee transactions and the SEQUENTIAL
part of HashInputs (message, SHA-256 chain: one lane per batch). The bit-level witness of the chain's
blocks -- 0.73 GB of the 0.76 GB the tail writes at (2048, 32) -- is independent per block given the
message and the chaining values, so rank 0 broadcasts those (96 B per block, 73 KB: the second and
last collective) and every rank expands blocks hz_shard_range(n_blocks, world, r) (SURVEY 8e "scatter
blocks back: 766 / 8"). The witness stays sharded in HBM: rank r holds the signals of its transactions
and of its SHA-256 blocks, rank 0 in addition the fee-tx section and the public output.
"""


def shard_ranges(lib, n_tx, world):
    return [lib.shard_range(n_tx, world, r) for r in range(world)]


def max_shard(ranges):
    return max(c for _, c in ranges)


class ShardedBatch:
    """Drives one sharded witness pass. `ctx` is a RollupMain Ctx (or any object with the same
    set_shard/enqueue/da_export/da_import/enqueue_tail[_chain]/sha_*/check methods); `alloc(nbytes)` returns a
    device byte buffer object with .data_ptr(), COMPLETE when it returns (a fill still queued on another stream would land on top of
    the first export: the pass runs on the caller's stream only); `all_gather(out_list_buf, in_buf)` gathers equal
    sized buffers from all ranks into rank order; `broadcast(buf)` (optional) sends rank 0's buffer to
    every rank -- with it the SHA-256 block witness is split over the ranks, without it rank 0 writes all of it."""

    def __init__(self, ctx, lib, n_tx, rank, world, alloc, all_gather, broadcast=None, force_split=False):
        self.ctx, self.rank, self.world, self.all_gather, self.broadcast = ctx, rank, world, all_gather, broadcast
        self.ranges = shard_ranges(lib, n_tx, world)
        self.first, self.count = self.ranges[rank]
        self.rec = ctx.da_record_bytes()
        self.slot = max_shard(self.ranges) * self.rec
        self.send = alloc(self.slot)
        self.recv = alloc(self.slot * world)
        self.split_tail = broadcast is not None and (world > 1 or force_split)   # force_split: the one-rank test of the collective path
        if self.split_tail:
            self.blocks = lib.shard_range(ctx.sha_blocks(), world, rank)
            self.sha = alloc(ctx.sha_state_bytes())
        ctx.set_shard(self.first, self.count, rank == 0)

    def step(self, stream, mark=None):
        """One sharded pass on HIP stream `stream` (a hipStream_t handle, required): export, collectives, imports and the tail
        are ordered by that stream, so `all_gather` / `broadcast` must enqueue on it too (bench.py: `with torch.cuda.stream(s)`),
        or block until their buffers are complete. `mark(label)` (optional) is called between the phases -- the caller records an
        event on the stream there (bench.py: per-rank phase times of the sharded line)."""
        mark = mark or (lambda label: None)
        if not stream:
            raise ValueError("ShardedBatch.step needs an explicit stream: the collective has to be ordered with the export/import kernels")
        c = self.ctx
        mark("start")
        c.enqueue(stream)                       # this rank's transactions
        c.da_export(self.send.data_ptr(), stream)
        mark("shard")
        self.all_gather(self.recv, self.send)   # collective 1: data-availability records
        mark("all_gather")
        if self.rank == 0:
            for r in range(1, self.world):
                f, n = self.ranges[r]
                c.da_import(f, n, self.recv.data_ptr() + r * self.slot, stream)
            if self.split_tail:
                c.enqueue_tail_chain(stream)    # FeeTx + message + the sequential SHA-256 chain
                c.sha_export(self.sha.data_ptr(), stream)
            else:
                c.enqueue_tail(stream)          # FeeTx + HashInputs, block witness included
        mark("tail")
        if self.split_tail:
            self.broadcast(self.sha)            # collective 2: message blocks and chaining values from rank 0
            mark("broadcast")
            f, n = self.blocks
            c.sha_expand(f, n, None if self.rank == 0 else self.sha.data_ptr(), stream)
            mark("expand")
        c.check()
