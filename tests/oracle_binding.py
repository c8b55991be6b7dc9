"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so) -- test infrastructure only."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def fr_to_bytes(vals):
    return b"".join(int(v % P).to_bytes(32, "little") for v in vals)


def fr_from_bytes(buf):
    buf = bytes(buf)
    return [int.from_bytes(buf[i:i + 32], "little") for i in range(0, len(buf), 32)]


class Oracle:
    def __init__(self):
        so = os.environ.get("ORACLE_SO") or os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.c = ctypes.CDLL(so)
        self.c.orc_poseidon_batch.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]

    def poseidon_batch(self, t, inputs, witness=False):
        n = len(inputs)
        flat = fr_to_bytes([x for row in inputs for x in row])
        out = ctypes.create_string_buffer(32 * max(n, 1))
        nsbox = 8 * t + [56, 57, 56, 60, 60, 63][t - 2]
        wit = ctypes.create_string_buffer(96 * nsbox * max(n, 1)) if witness else None
        rc = self.c.orc_poseidon_batch(t, n, flat, out, wit)
        assert rc == 0
        return fr_from_bytes(out.raw[:32 * n]), (wit.raw if witness else None)


TEMPLATES = {"rollup-main": 0, "rollup-tx": 1, "decode-tx": 2, "fee-tx": 3, "hash-state": 4, "withdraw": 5, "hash-inputs": 6, "decode-float": 7, "compute-fee": 8, "fee-accumulator": 9, "balance-updater": 10,
             "rollup-tx-states": 11, "rq-tx-verifier": 12, "mux256": 13, "bits-compressed-2-ay-sign": 14, "ay-sign-2-ax": 15, "smt-processor": 16, "smt-verifier": 17}


def flatten(v):
    if isinstance(v, (list, tuple)):
        out = []
        for x in v:
            out.extend(flatten(x))
        return out
    return [int(v)]


class OracleCtx:
    """One oracle "circuit": mirrors the product's hz_ctx API (oracle/oracle_api.h)."""

    def __init__(self, template, nTx=0, nLevels=0, maxL1Tx=0, maxFeeTx=0, n_instances=1):
        self.o = Oracle()
        c = self.o.c
        c.orc_ctx_create.restype = ctypes.c_void_p
        c.orc_ctx_destroy.argtypes = [ctypes.c_void_p]
        c.orc_witness_len.restype = ctypes.c_uint64
        c.orc_witness_len.argtypes = [ctypes.c_void_p]
        c.orc_witness_total.restype = ctypes.c_uint64
        c.orc_witness_total.argtypes = [ctypes.c_void_p]
        c.orc_set_input.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        c.orc_run.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
        c.orc_read.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        c.orc_read_raw.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
        c.orc_unwritten.restype = ctypes.c_uint64
        c.orc_unwritten.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        c.orc_symbol_lookup.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
        c.orc_symbol_count.restype = ctypes.c_uint64
        c.orc_symbol_count.argtypes = [ctypes.c_void_p]
        c.orc_constraint_name.restype = ctypes.c_char_p
        self.h = c.orc_ctx_create(TEMPLATES[template], nTx, nLevels, maxL1Tx, maxFeeTx, n_instances)
        self.n_instances = n_instances

    def __del__(self):
        try:
            self.o.c.orc_ctx_destroy(self.h)
        except Exception:
            pass

    def witness_len(self):
        return self.o.c.orc_witness_len(self.h)

    def total(self):
        return self.o.c.orc_witness_total(self.h)

    def set_input(self, name, value, instance=0):
        flat = flatten(value)
        rc = self.o.c.orc_set_input(self.h, instance, name.encode(), fr_to_bytes(flat), len(flat))
        if rc:
            raise ValueError("oracle rejected input %s (rc %d, %d values)" % (name, rc, len(flat)))

    def set_inputs(self, d, instance=0):
        for k, v in d.items():
            self.set_input(k, v, instance)

    def run(self):
        """returns None or (instance, unit, cid, name, lhs, rhs)"""
        a, b, cc = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        l, r = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        rc = self.o.c.orc_run(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(cc), l, r)
        if rc == 0:
            return None
        if rc == 3:
            return (a.value, b.value, cc.value, self.o.c.orc_constraint_name(cc.value).decode(), int.from_bytes(l.raw, "little"), int.from_bytes(r.raw, "little"))
        raise RuntimeError("oracle run rc=%d (missing input?)" % rc)

    def failure_of(self, instance):
        """first failure of one instance in the last run: None or (unit, constraint id, lhs, rhs)"""
        u, cc = ctypes.c_int32(), ctypes.c_int32()
        l, r = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        f = self.o.c.orc_failure_of
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, ctypes.c_void_p]
        rc = f(self.h, instance, ctypes.byref(u), ctypes.byref(cc), l, r)
        if rc == 0:
            return None
        assert rc == 3
        return (u.value, cc.value, int.from_bytes(l.raw, "little"), int.from_bytes(r.raw, "little"))

    def read(self, first, count, instance=0):
        buf = ctypes.create_string_buffer(32 * count)
        assert self.o.c.orc_read(self.h, instance, first, count, buf) == 0
        return fr_from_bytes(buf.raw)

    def log_poseidon(self, on=True):
        """keep the inputs of every Poseidon component of the next runs (poseidon_inputs)"""
        self.o.c.orc_log_poseidon.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.o.c.orc_log_poseidon(self.h, 1 if on else 0)

    def poseidon_inputs(self, virt_first, instance=0):
        """inputs the Poseidon component whose first stored signal has per-instance index `virt_first` was evaluated on, or None"""
        f = self.o.c.orc_poseidon_inputs
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int]
        buf = ctypes.create_string_buffer(32 * 8)
        n = f(self.h, instance, virt_first, buf, 8)
        return fr_from_bytes(buf.raw[:32 * n]) if n else None

    def read_bytes(self, first, count, instance=0):
        buf = ctypes.create_string_buffer(32 * count)
        assert self.o.c.orc_read(self.h, instance, first, count, buf) == 0
        return buf.raw

    def read_raw_bytes(self, first=0, count=None):
        count = self.total() - first if count is None else count
        buf = ctypes.create_string_buffer(32 * count)
        assert self.o.c.orc_read_raw(self.h, first, count, buf) == 0
        return buf.raw

    def lookup(self, name):
        idx = ctypes.c_uint64()
        if not self.o.c.orc_symbol_lookup(self.h, name.encode(), ctypes.byref(idx)):
            raise KeyError(name)
        return idx.value

    def get(self, name, instance=0):
        return self.read(self.lookup(name), 1, instance)[0]

    def symbol_names(self):
        """names of all stored signals of one instance (Poseidon blocks as '<component>.sigma*')"""
        f = self.o.c.orc_symbol_names
        f.restype = ctypes.c_uint64
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        n = f(self.h, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        f(self.h, buf, n)
        return buf.raw[:n].decode().split("\n")[:-1]

    def unwritten(self):
        nm = ctypes.create_string_buffer(256)
        n = self.o.c.orc_unwritten(self.h, nm, 256)
        return n, nm.value.decode()
