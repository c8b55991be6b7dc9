"""ctypes view of the NATIVE batch builder of circuits_amd/libhz_host.so (include/hz_host.h: hzb_db_* / hzb_batch_*): the same
RollupDB / BatchBuilder interface as circuits_amd/builder.py -- the role @hermeznetwork/commonjs plays for the reference
(test/helpers/helpers.js:46,148, tools/generate-input.js:70-107) -- with the transaction walk, the Merkle bookkeeping and the
signing of synthetic transactions in C++ and the hashing handed to hz_poseidon_dag as ONE DAG per batch. A batch comes back as
the packed bulk-upload buffer of hz_inputs_upload plus the expected hashGlobalInputs; no Python object per signal is built.

Caller-side code: circuit INPUTS only, nothing from oracle/."""
import ctypes
import os

from . import builder as B

_HERE = os.path.dirname(os.path.abspath(__file__))

HAS_AUX_TO, HAS_NONCE, HAS_RQ, HAS_SIG, HAS_SIGNER = 1, 2, 4, 8, 16


class hzb_leaf(ctypes.Structure):
    _fields_ = [("token_id", ctypes.c_uint32), ("sign", ctypes.c_uint32), ("nonce", ctypes.c_uint64),
                ("balance", ctypes.c_uint8 * 32), ("ay", ctypes.c_uint8 * 32), ("eth_addr", ctypes.c_uint8 * 32)]


class hzb_tx(ctypes.Structure):
    _fields_ = [("from_idx", ctypes.c_uint64), ("to_idx", ctypes.c_uint64), ("aux_to_idx", ctypes.c_uint64),
                ("amount_f", ctypes.c_uint64), ("load_amount_f", ctypes.c_uint64), ("nonce", ctypes.c_uint64),
                ("token_id", ctypes.c_uint32), ("max_num_batch", ctypes.c_uint32),
                ("on_chain", ctypes.c_uint8), ("user_fee", ctypes.c_uint8), ("rq_offset", ctypes.c_uint8), ("to_bjj_sign", ctypes.c_uint8),
                ("flags", ctypes.c_uint32)] + [(n, ctypes.c_uint8 * 32) for n in (
                    "to_eth_addr", "to_bjj_ay", "from_eth_addr", "from_bjj_compressed", "rq_tx_compressed_data_v2", "rq_to_eth_addr", "rq_to_bjj_ay",
                    "r8x", "r8y", "s", "signer_key")]


_tx_dtype = None


def tx_dtype():
    """hzb_tx as a numpy record type (same field names, offsets and size): synthetic batches fill it by columns"""
    global _tx_dtype
    if _tx_dtype is None:
        import numpy as np
        fields = []
        for name, ct in hzb_tx._fields_:
            fields.append((name, "V32") if ctypes.sizeof(ct) == 32 else (name, {8: "<u8", 4: "<u4", 1: "u1"}[ctypes.sizeof(ct)]))
        dt = np.dtype(fields, align=True)
        assert dt.itemsize == ctypes.sizeof(hzb_tx) and all(dt.fields[n][1] == getattr(hzb_tx, n).offset for n, _ in hzb_tx._fields_)
        _tx_dtype = dt
    return _tx_dtype


DAG_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p)

_lib = None


def host_lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libhz_host.so")
        if not os.path.exists(path):
            raise RuntimeError("%s missing: run __graft_entry__.build()" % path)
        c = ctypes.CDLL(path)
        c.hzb_last_error.restype = ctypes.c_char_p
        c.hzb_db_create.restype = ctypes.c_void_p
        c.hzb_db_create.argtypes = [ctypes.c_uint32, ctypes.c_uint64]
        c.hzb_db_destroy.argtypes = [ctypes.c_void_p]
        c.hzb_db_clone.restype = ctypes.c_void_p
        c.hzb_db_clone.argtypes = [ctypes.c_void_p]
        c.hzb_db_set_dag.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        c.hzb_db_set_base.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64] + [ctypes.c_void_p] * 5 + [ctypes.c_int32] + [ctypes.c_void_p] * 3
        c.hzb_db_add_account.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        c.hzb_db_get_account.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        c.hzb_db_state_root.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        c.hzb_db_last_idx.restype = ctypes.c_uint64
        c.hzb_db_last_idx.argtypes = [ctypes.c_void_p]
        c.hzb_db_num_batch.restype = ctypes.c_uint32
        c.hzb_db_num_batch.argtypes = [ctypes.c_void_p]
        c.hzb_batch_create.restype = ctypes.c_void_p
        c.hzb_batch_create.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 4
        c.hzb_batch_destroy.argtypes = [ctypes.c_void_p]
        c.hzb_batch_add_tx.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        c.hzb_batch_add_txs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        c.hzb_batch_add_synthetic.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_char_p]
        c.hzb_batch_add_token.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        c.hzb_batch_add_fee_idx.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        c.hzb_batch_build.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        c.hzb_batch_build_begin.argtypes = c.hzb_batch_build.argtypes
        c.hzb_batch_build_finish.argtypes = [ctypes.c_void_p]
        c.hzb_batch_roots.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 3
        c.hzb_batch_exit_proof.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        c.hzb_batch_tx_flags.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        c.hzb_batch_stats.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 5
        _lib = c
    return _lib


class BuilderError(ValueError):
    """what circuits_amd/builder.py raises as ValueError / KeyError: bad arguments, or a transaction the circuit would reject"""

    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


def _check(st):
    if st:
        raise BuilderError(st, host_lib().hzb_last_error().decode())


_ZERO32 = bytes(32)


def _b32(v):
    return (ctypes.c_uint8 * 32).from_buffer_copy(int(v).to_bytes(32, "little"))


def _int(arr):
    return int.from_bytes(bytes(arr), "little")


def _leaf_struct(st):
    return hzb_leaf(st["tokenID"], st["sign"], st["nonce"], _b32(st["balance"]), _b32(st["ay"]), _b32(st["ethAddr"]))


def _leaf_dict(lf):
    return {"tokenID": lf.token_id, "nonce": lf.nonce, "sign": lf.sign, "balance": _int(lf.balance), "ay": _int(lf.ay), "ethAddr": _int(lf.eth_addr)}


def tx_struct(tx):
    """a transaction object of the reference's suites (dict, as circuits_amd/builder.py BatchBuilder.add_tx takes it) -> hzb_tx"""
    t = hzb_tx()
    flags = 0
    t.from_idx, t.to_idx = tx.get("fromIdx", 0), tx.get("toIdx", 0)
    if "auxToIdx" in tx:
        t.aux_to_idx, flags = tx["auxToIdx"], flags | HAS_AUX_TO
    t.amount_f = B.fix2float(tx.get("amount", 0)) if ("amountF" not in tx or "amount" in tx) else tx["amountF"]
    t.load_amount_f = tx.get("loadAmountF", 0)
    if "nonce" in tx:
        t.nonce, flags = tx["nonce"], flags | HAS_NONCE
    t.token_id, t.max_num_batch = tx.get("tokenID", 0), tx.get("maxNumBatch", 0)
    t.on_chain, t.user_fee, t.rq_offset, t.to_bjj_sign = (1 if tx.get("onChain") else 0), tx.get("userFee", 0), tx.get("rqOffset", 0), tx.get("toBjjSign", 0)
    t.to_eth_addr, t.to_bjj_ay = _b32(tx.get("toEthAddr", 0)), _b32(tx.get("toBjjAy", 0))
    t.from_eth_addr, t.from_bjj_compressed = _b32(tx.get("fromEthAddr", 0)), _b32(tx.get("fromBjjCompressed", 0))
    if "rqTxCompressedDataV2" in tx:
        flags |= HAS_RQ
        t.rq_tx_compressed_data_v2, t.rq_to_eth_addr, t.rq_to_bjj_ay = _b32(tx["rqTxCompressedDataV2"]), _b32(tx.get("rqToEthAddr", 0)), _b32(tx.get("rqToBjjAy", 0))
    if "signer" in tx:
        flags |= HAS_SIGNER
        t.signer_key = _b32(tx["signer"].k)
    elif any(k in tx for k in ("r8x", "r8y", "s")):
        flags |= HAS_SIG
        t.r8x, t.r8y, t.s = _b32(tx.get("r8x", 0)), _b32(tx.get("r8y", 0)), _b32(tx.get("s", 0))
    t.flags = flags
    return t


class NativeRollupDB:
    """RollupDB of circuits_amd/builder.py on the native library. device=N: the batch's hashes run on GPU N through hz_poseidon_dag
    (no CPU fallback there: a missing device is an error); device=None: the library's host Poseidon. dag_fn: any function with
    hz_poseidon_dag's signature (tests)."""

    def __init__(self, chain_id=1, device=None, first_idx=256, base=None, dag_fn=None):
        self.c = host_lib()
        self.h = ctypes.c_void_p(self.c.hzb_db_create(chain_id, first_idx))
        self.chain_id = chain_id
        self._keep = []
        if device is not None:
            from . import lib
            L = lib()
            if L.device_count() <= 0:
                raise RuntimeError("NativeRollupDB(device=%d): no usable gfx950 device (the device batch builder has no CPU fallback)" % device)
            _check(self.c.hzb_db_set_dag(self.h, ctypes.cast(L.c.hz_poseidon_dag, ctypes.c_void_p), device))
        elif dag_fn is not None:
            self._keep.append(dag_fn)
            _check(self.c.hzb_db_set_dag(self.h, ctypes.cast(dag_fn, ctypes.c_void_p), 0))
        if base is not None:
            self.set_base(base)

    def close(self):
        if self.h:
            self.c.hzb_db_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clone(self):
        """a working copy: build a batch on it, keep it (consolidate) or drop it"""
        c = object.__new__(NativeRollupDB)
        c.c, c.chain_id, c._keep = self.c, self.chain_id, self._keep
        h = self.c.hzb_db_clone(self.h)
        if not h:
            raise BuilderError(1, self.c.hzb_last_error().decode())
        c.h = ctypes.c_void_p(h)
        return c

    def set_base(self, base):
        """a builder.DenseState as the pre-populated part of the state; its arrays are used in place"""
        import numpy as np
        lv = [np.ascontiguousarray(a, dtype=np.uint8) for a in base.levels]
        value = np.ascontiguousarray(base.value, dtype=np.uint8)
        key_idx = np.ascontiguousarray(base.key_idx, dtype=np.uint8)
        mant = np.ascontiguousarray(base.mant, dtype=np.uint64)
        expo = np.ascontiguousarray(base.expo, dtype=np.uint8)
        keys = base.keys()
        sign = np.array([a.sign for a in keys], dtype=np.uint8)
        ay = np.frombuffer(b"".join(a.ay.to_bytes(32, "little") for a in keys), dtype=np.uint8).copy()
        eth = np.frombuffer(b"".join(a.eth_addr.to_bytes(32, "little") for a in keys), dtype=np.uint8).copy()
        ptrs = (ctypes.c_void_p * len(lv))(*[a.ctypes.data for a in lv])
        self._keep += [lv, value, key_idx, mant, expo, sign, ay, eth, ptrs]
        _check(self.c.hzb_db_set_base(self.h, base.k, base.first_idx, ptrs, value.ctypes.data, key_idx.ctypes.data, mant.ctypes.data, expo.ctypes.data,
                                      len(keys), sign.ctypes.data, ay.ctypes.data, eth.ctypes.data))

    def add_account(self, st):
        idx = ctypes.c_uint64()
        lf = _leaf_struct(st)
        _check(self.c.hzb_db_add_account(self.h, ctypes.byref(lf), ctypes.byref(idx)))
        return idx.value

    def account(self, idx):
        lf = hzb_leaf()
        _check(self.c.hzb_db_get_account(self.h, idx, ctypes.byref(lf)))
        return _leaf_dict(lf)

    @property
    def state_root(self):
        out = (ctypes.c_uint8 * 32)()
        _check(self.c.hzb_db_state_root(self.h, out))
        return _int(out)

    @property
    def last_idx(self):
        return self.c.hzb_db_last_idx(self.h)

    @property
    def num_batch(self):
        return self.c.hzb_db_num_batch(self.h)

    def build_batch(self, n_tx, n_levels, max_l1, max_fee):
        return NativeBatchBuilder(self, n_tx, n_levels, max_l1, max_fee)


def layout_tables(layout):
    """Ctx.packed_layout() -> the (count, names, offsets, widths, total) arrays hzb_batch_build takes; build once per circuit shape"""
    total, sigs = layout
    n = len(sigs)
    names = (ctypes.c_char_p * n)(*[s[0].encode() for s in sigs])
    offs = (ctypes.c_uint64 * n)(*[s[1] for s in sigs])
    widths = (ctypes.c_uint32 * n)(*[s[2] for s in sigs])
    return n, names, offs, widths, total


class NativeBatchBuilder:
    def __init__(self, db, n_tx, n_levels, max_l1, max_fee):
        self.db, self.nTx, self.L, self.maxL1, self.F = db, n_tx, n_levels, max_l1, max_fee
        self.c = db.c
        h = self.c.hzb_batch_create(db.h, n_tx, n_levels, max_l1, max_fee)
        if not h:
            raise BuilderError(1, self.c.hzb_last_error().decode())
        self.h = ctypes.c_void_p(h)
        self.hash_global_inputs = None

    def close(self):
        if self.h:
            self.c.hzb_batch_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_tx(self, tx):
        t = tx if isinstance(tx, hzb_tx) else tx_struct(tx)
        _check(self.c.hzb_batch_add_tx(self.h, ctypes.byref(t)))

    def add_txs(self, arr):
        """hzb_batch_add_txs: a numpy array of TX_DTYPE records (the hzb_tx layout), all in one call"""
        assert arr.dtype == tx_dtype() and arr.flags.c_contiguous
        _check(self.c.hzb_batch_add_txs(self.h, arr.ctypes.data, len(arr)))

    def add_token(self, token_id):
        _check(self.c.hzb_batch_add_token(self.h, token_id))

    def add_fee_idx(self, idx):
        _check(self.c.hzb_batch_add_fee_idx(self.h, idx))

    def build(self, layout, out=None):
        """layout: Ctx.packed_layout() or layout_tables(...) of it. out: address of a (pinned) buffer of `total` bytes, or None for a
        bytes object. Returns (packed, hashGlobalInputs)."""
        n, names, offs, widths, total = layout if len(layout) == 5 else layout_tables(layout)
        buf = None
        if out is None:
            buf = ctypes.create_string_buffer(total)
            out = ctypes.addressof(buf)
        else:
            ctypes.memset(out, 0, total)
        hgi = (ctypes.c_uint8 * 32)()
        _check(self.c.hzb_batch_build(self.h, n, names, offs, widths, out, total, hgi))
        self.hash_global_inputs = _int(hgi)
        self._built = ((buf.raw if buf is not None else out), self.hash_global_inputs)   # build_finish() after build(): the same pair
        return self._built

    def build_begin(self, layout, out=None):
        """first half of build() (hzb_batch_build_begin): the walk; the batch's Merkle hashes are evaluated by a worker thread while the
        caller goes on -- typically to the next batch, on this database or another. build_finish() completes it."""
        n, names, offs, widths, total = layout if len(layout) == 5 else layout_tables(layout)
        self._buf = None
        if out is None:
            self._buf = ctypes.create_string_buffer(total)
            out = ctypes.addressof(self._buf)
        else:
            ctypes.memset(out, 0, total)
        self._out, self._hgi = out, (ctypes.c_uint8 * 32)()
        _check(self.c.hzb_batch_build_begin(self.h, n, names, offs, widths, out, total, self._hgi))

    def build_finish(self):
        """-> (packed, hashGlobalInputs) as build()"""
        if getattr(self, "_built", None) is not None:
            return self._built
        _check(self.c.hzb_batch_build_finish(self.h))
        self.hash_global_inputs = _int(self._hgi)
        return (self._buf.raw if self._buf is not None else self._out), self.hash_global_inputs

    def get_hash_inputs(self):
        return self.hash_global_inputs

    def roots(self):
        a, b, li = (ctypes.c_uint8 * 32)(), (ctypes.c_uint8 * 32)(), ctypes.c_uint64()
        _check(self.c.hzb_batch_roots(self.h, a, b, ctypes.byref(li)))
        return _int(a), _int(b), li.value

    def exit_proof(self, idx):
        """(leaf dict, siblings padded to nLevels + 1) of an exit leaf: what Withdraw(nLevels) takes (reference test/withdraw.test.js:39-157)"""
        lf, sib, n = hzb_leaf(), (ctypes.c_uint8 * (32 * (self.L + 1)))(), ctypes.c_int32()
        _check(self.c.hzb_batch_exit_proof(self.h, idx, ctypes.byref(lf), sib, ctypes.byref(n)))
        raw = bytes(sib)
        return _leaf_dict(lf), [int.from_bytes(raw[32 * k:32 * k + 32], "little") for k in range(self.L + 1)]

    def is_amount_nullified(self, i):
        f = ctypes.c_int32()
        _check(self.c.hzb_batch_tx_flags(self.h, i, ctypes.byref(f)))
        return f.value

    def stats(self):
        jobs, segs = ctypes.c_uint64(), ctypes.c_uint64()
        dms, walk, ev = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _check(self.c.hzb_batch_stats(self.h, ctypes.byref(jobs), ctypes.byref(segs), ctypes.byref(dms), ctypes.byref(walk), ctypes.byref(ev)))
        f = self.c.hzb_batch_sign_s
        f.restype, f.argtypes = ctypes.c_double, [ctypes.c_void_p]
        return {"jobs": jobs.value, "segments": segs.value, "device_ms": dms.value, "walk_s": walk.value, "eval_s": ev.value, "sign_s": f(self.h)}


def synthetic_batch_native(n_tx, n_levels, max_l1, max_fee, layout, seed=0x48455A31, n_accounts=None, n_keys=8, exits=0, device=None, first_idx=256,
                           base=None, out=None, native_recipe=False, begin_only=False, phases=None):
    """builder.synthetic_batch's recipe (reference tools/generate-input.js:61-109) on the native builder: the same seeded
    transactions, hence the same circuit inputs byte for byte. Pre-population goes through a DenseState (built here when `base` is None
    and n_accounts is a power of two >= 16, as synthetic_batch(dense=True) does). Returns (batch, packed, hashGlobalInputs)."""
    import random
    rng = random.Random(seed)
    if base is None:
        n_accounts = n_accounts if n_accounts is not None else max(2, min(4 * n_tx, 4096))
        if not (n_accounts >= 16 and n_accounts & (n_accounts - 1) == 0):
            raise ValueError("synthetic_batch_native: n_accounts must be a power of two >= 16 (DenseState)")
        base = B.DenseState.build(n_accounts.bit_length() - 1, seed=seed, first_idx=first_idx, n_keys=n_keys)
    import numpy as np
    import time
    tick = [time.perf_counter()]

    def phase(name):   # where the time of a build goes besides the walk (bench.py: config.batch_builder.phases_ms)
        t = time.perf_counter()
        if phases is not None:
            phases[name] = phases.get(name, 0.0) + t - tick[0]
        tick[0] = t
    db = NativeRollupDB(chain_id=1, device=device, base=base)
    phase("database")
    keys = [B.Account(seed * 1000 + i) for i in range(n_keys)]
    bkeys = base.keys()
    phase("l1_keys")
    bb = db.build_batch(n_tx, n_levels, max_l1, max_fee)
    n_l1 = min(max_l1, n_tx)
    if native_recipe:
        # the whole recipe below inside libhz_host.so (hzb_batch_add_synthetic: CPython's generator restated bit for bit): no Python per
        # transaction between the seed and the packed inputs
        l1b = b"".join(a.bjj_compressed.to_bytes(32, "little") for a in keys)
        l1e = b"".join(a.eth_addr.to_bytes(32, "little") for a in keys)
        sk = b"".join(a.k.to_bytes(32, "little") for a in bkeys)
        _check(bb.c.hzb_batch_add_synthetic(bb.h, ctypes.c_uint64(seed), exits, n_keys, l1b, l1e, len(bkeys), sk))
        phase("recipe")
        bb._db_keep = db
        if begin_only:   # the caller finishes it (build_finish) after it has begun the next batch
            bb.build_begin(layout, out)
            phase("build_begin")
            return bb
        packed, hgi = bb.build(layout, out)
        return bb, packed, hgi
    # the transactions as columns of one hzb_tx array (one hzb_batch_add_txs call); 32-byte fields as rows of bytes
    col = {k: [0] * n_tx for k in ("from_idx", "to_idx", "amount_f", "load_amount_f", "nonce", "user_fee", "on_chain", "flags")}
    wide = {k: [_ZERO32] * n_tx for k in ("from_eth_addr", "from_bjj_compressed", "signer_key")}
    l1_fields = [(a.bjj_compressed.to_bytes(32, "little"), a.eth_addr.to_bytes(32, "little")) for a in keys]
    for i in range(n_l1):
        q = rng.randrange(n_keys)
        col["load_amount_f"][i] = B.floor_fix2float(rng.randrange(1 << 96))
        col["on_chain"][i] = 1
        wide["from_bjj_compressed"][i], wide["from_eth_addr"][i] = l1_fields[q]
    tmp = {}
    pick = lambda: base.first_idx + rng.randrange(base.N)   # noqa: E731
    signer_bytes = [a.k.to_bytes(32, "little") for a in bkeys]
    for t in range(n_tx - n_l1):
        i = n_l1 + t
        frm, to = pick(), pick()
        if frm in tmp:
            bal, nonce = tmp[frm]
        else:
            st = base.state(frm)
            bal, nonce = st["balance"], st["nonce"]
        amount_f = B.floor_fix2float(bal * 20 // 100)
        amount = B.float2fix(amount_f)
        is_exit = t < exits
        col["from_idx"][i], col["to_idx"][i], col["amount_f"][i], col["nonce"][i] = frm, (B.EXIT_IDX if is_exit else to), amount_f, nonce
        col["user_fee"][i], col["flags"][i] = 176, HAS_NONCE | HAS_SIGNER
        wide["signer_key"][i] = signer_bytes[int(base.key_idx[frm - base.first_idx])]
        nb = bal - amount - B.compute_fee(amount, 176)
        tmp[frm] = (nb, nonce + 1)
        if not is_exit and to != frm:
            if to in tmp:
                tb, tn = tmp[to]
            else:
                st = base.state(to)
                tb, tn = st["balance"], st["nonce"]
            tmp[to] = (tb + amount, tn)
        elif not is_exit and to == frm:
            tmp[frm] = (nb + amount, nonce + 1)
    arr = np.zeros(n_tx, dtype=tx_dtype())
    for k, v in col.items():
        arr[k] = v
    arr["token_id"] = 1
    for k, v in wide.items():
        arr[k] = np.frombuffer(b"".join(v), dtype="V32")
    bb.add_txs(arr)
    bb.add_token(1)
    bb.add_fee_idx(pick())
    bb._db_keep = db
    if begin_only:
        bb.build_begin(layout, out)
        return bb
    packed, hgi = bb.build(layout, out)
    return bb, packed, hgi
