"""ctypes binding of include/hermez_witness.h (no torch types cross this boundary)."""
import ctypes
import os

R_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    # HZ_WITNESS_LIB: alternative build of the same library (kernel tuning experiments)
    return os.environ.get("HZ_WITNESS_LIB") or os.path.join(_HERE, "libhermez_witness.so")


class HzError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("hz status %d: %s" % (status, msg))
        self.status = status


class ConstraintError(HzError):
    """A `===` of the circuit failed; message mirrors circom_runtime's text."""

    def __init__(self, instance, unit, cid, name, lhs, rhs):
        RuntimeError.__init__(self, "Constraint doesn't match %d != %d (%s, instance %d unit %d)" % (lhs, rhs, name, instance, unit))
        self.status = 3
        self.instance, self.unit, self.constraint_id, self.name, self.lhs, self.rhs = instance, unit, cid, name, lhs, rhs


def fr_to_bytes(vals):
    return b"".join(int(v % R_MODULUS).to_bytes(32, "little") for v in vals)


def fr_from_bytes(buf):
    buf = bytes(buf)
    return [int.from_bytes(buf[i:i + 32], "little") for i in range(0, len(buf), 32)]


class hz_params(ctypes.Structure):
    _fields_ = [("template_id", ctypes.c_int32), ("nTx", ctypes.c_int32), ("nLevels", ctypes.c_int32),
                ("maxL1Tx", ctypes.c_int32), ("maxFeeTx", ctypes.c_int32), ("device", ctypes.c_int32),
                ("n_instances", ctypes.c_int32), ("flags", ctypes.c_int32)]


class hz_error(ctypes.Structure):
    _fields_ = [("instance", ctypes.c_int32), ("unit", ctypes.c_int32), ("constraint_id", ctypes.c_int32),
                ("lhs", ctypes.c_uint8 * 32), ("rhs", ctypes.c_uint8 * 32)]


class hz_symbol(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("index", ctypes.c_uint64)]


TEMPLATES = {"rollup-main": 0, "rollup-tx": 1, "decode-tx": 2, "fee-tx": 3, "hash-state": 4, "withdraw": 5, "hash-inputs": 6, "decode-float": 7, "compute-fee": 8, "fee-accumulator": 9, "balance-updater": 10,
             "rollup-tx-states": 11, "rq-tx-verifier": 12, "mux256": 13, "bits-compressed-2-ay-sign": 14, "ay-sign-2-ax": 15, "smt-processor": 16, "smt-verifier": 17}

# every symbol include/hermez_witness.h declares; tests check the .so exports all of them
EXPORTS = [
    "hz_version", "hz_last_error", "hz_device_count", "hz_ctx_create", "hz_ctx_destroy", "hz_witness_len", "hz_ctx_device_bytes", "hz_template_device_bytes",
    "hz_constraint_estimate", "hz_set_input", "hz_set_input_dev", "hz_copy_instance_inputs", "hz_inputs_packed_bytes", "hz_input_packed_width",
    "hz_input_packed_offset", "hz_host_alloc", "hz_host_free", "hz_inputs_upload", "hz_inputs_stage", "hz_inputs_stage_range", "hz_clear_inputs", "hz_input_count", "hz_input_name",
    "hz_witness_enqueue", "hz_witness_check", "hz_witness_run", "hz_witness_failures", "hz_witness_read", "hz_witness_dev_ptr",
    "hz_witness_total", "hz_witness_read_raw", "hz_ctx_set_profiling", "hz_profile_count", "hz_profile_get",
    "hz_ctx_set_shard", "hz_da_record_bytes", "hz_da_export", "hz_da_import", "hz_witness_enqueue_tail",
    "hz_witness_enqueue_tail_chain", "hz_sha_blocks", "hz_sha_state_bytes", "hz_sha_export", "hz_sha_expand",
    "hz_comm_create", "hz_comm_destroy", "hz_comm_rank", "hz_comm_world", "hz_shard_step", "hz_ctx_ntx",
    "hz_symmap_create", "hz_symmap_create_r1cs", "hz_symmap_solved", "hz_symmap_check_r1cs", "hz_symmap_save", "hz_symmap_load", "hz_symmap_destroy", "hz_symmap_nvars", "hz_symmap_unresolved", "hz_symmap_derived", "hz_witness_read_sym", "hz_witness_write_wtns_sym", "hz_witness_gather",
    "hz_symmap_from_index", "hz_component_major_index", "hz_symmap_upload", "hz_witness_export_dev", "hz_witness_export_range_dev", "hz_witness_export_host", "hz_symmap_dev_index", "hz_witness_derive_dev",
    "hz_symbol_count", "hz_symbol_get", "hz_symbol_lookup", "hz_constraint_name", "hz_poseidon_batch",
    "hz_poseidon_batch_dev", "hz_shard_range", "hz_set_inputs_json", "hz_witness_write_json", "hz_witness_write_wtns", "hz_symbols_write_sym", "hz_fr_ops", "hz_poseidon_dag",
]


class Lib:
    """Loaded libhermez_witness.so. Raises if the HIP library has not been built."""

    def __init__(self, path=None):
        path = path or lib_path()
        if not os.path.exists(path):
            raise HzError(-1, "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        self.c = ctypes.CDLL(path)
        c = self.c
        c.hz_version.restype = ctypes.c_char_p
        c.hz_last_error.restype = ctypes.c_char_p
        c.hz_device_count.restype = ctypes.c_int32
        c.hz_poseidon_batch.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
        c.hz_poseidon_batch_dev.argtypes = [ctypes.c_int32, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        c.hz_poseidon_dag.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        c.hz_fr_ops.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        c.hz_shard_range.argtypes = [ctypes.c_int32] * 3 + [ctypes.POINTER(ctypes.c_int32)] * 2
        c.hz_shard_range.restype = None
        vp, u64 = ctypes.c_void_p, ctypes.c_uint64
        c.hz_ctx_create.argtypes = [ctypes.POINTER(hz_params), ctypes.POINTER(vp)]
        c.hz_ctx_destroy.argtypes = [vp]
        c.hz_ctx_destroy.restype = None
        for f in ("hz_witness_len", "hz_ctx_device_bytes", "hz_template_device_bytes", "hz_constraint_estimate", "hz_witness_total", "hz_symbol_count"):
            getattr(c, f).argtypes = [vp]
            getattr(c, f).restype = u64
        c.hz_set_input.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        c.hz_set_input_dev.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p, vp, ctypes.c_size_t, vp]
        c.hz_copy_instance_inputs.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp]
        c.hz_clear_inputs.argtypes = [vp]
        c.hz_clear_inputs.restype = None
        c.hz_inputs_packed_bytes.argtypes = [vp]
        c.hz_inputs_packed_bytes.restype = u64
        c.hz_input_packed_width.argtypes = [vp, ctypes.c_int32]
        c.hz_input_packed_offset.argtypes = [vp, ctypes.c_int32]
        c.hz_input_packed_offset.restype = u64
        c.hz_host_alloc.argtypes = [ctypes.c_size_t]
        c.hz_host_alloc.restype = vp
        c.hz_host_free.argtypes = [vp]
        c.hz_host_free.restype = None
        c.hz_inputs_upload.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_size_t, vp]
        c.hz_inputs_stage.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_size_t, vp]
        c.hz_inputs_stage_range.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_size_t, ctypes.c_size_t, vp]
        c.hz_input_count.argtypes = [vp]
        c.hz_input_name.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(u64)]
        c.hz_input_name.restype = ctypes.c_char_p
        c.hz_witness_enqueue.argtypes = [vp, vp]
        c.hz_witness_check.argtypes = [vp, ctypes.POINTER(hz_error)]
        c.hz_witness_run.argtypes = [vp, ctypes.POINTER(hz_error)]
        c.hz_witness_failures.argtypes = [vp, ctypes.POINTER(hz_error), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        c.hz_witness_read.argtypes = [vp, ctypes.c_int32, u64, u64, vp]
        c.hz_witness_read_raw.argtypes = [vp, u64, u64, vp]
        c.hz_witness_dev_ptr.argtypes = [vp]
        c.hz_witness_dev_ptr.restype = vp
        c.hz_set_inputs_json.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p, ctypes.c_size_t]
        c.hz_witness_write_json.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p]
        c.hz_witness_write_wtns.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p]
        c.hz_symbols_write_sym.argtypes = [vp, ctypes.c_char_p]
        c.hz_symbol_get.argtypes = [vp, u64, ctypes.POINTER(hz_symbol)]
        c.hz_symmap_create.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(vp)]
        c.hz_symmap_create_r1cs.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(vp)]
        c.hz_symmap_solved.argtypes = [vp]
        c.hz_symmap_solved.restype = u64
        c.hz_symmap_check_r1cs.argtypes = [vp, vp, ctypes.c_int32, ctypes.POINTER(u64), ctypes.POINTER(u64), u64]
        c.hz_symmap_save.argtypes = [vp, vp, ctypes.c_char_p]
        c.hz_symmap_load.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp)]
        c.hz_symmap_destroy.argtypes = [vp]
        c.hz_symmap_destroy.restype = None
        c.hz_symmap_nvars.argtypes = [vp]
        c.hz_symmap_nvars.restype = u64
        c.hz_symmap_unresolved.argtypes = [vp, u64, ctypes.POINTER(u64), ctypes.POINTER(ctypes.c_char_p)]
        c.hz_symmap_unresolved.restype = u64
        c.hz_witness_read_sym.argtypes = [vp, vp, ctypes.c_int32, u64, u64, vp]
        c.hz_witness_write_wtns_sym.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_char_p]
        c.hz_witness_gather.argtypes = [vp, ctypes.c_int32, vp, u64, vp]
        c.hz_symmap_upload.argtypes = [vp, vp, ctypes.POINTER(u64)]
        c.hz_symmap_from_index.argtypes = [vp, vp, u64, ctypes.POINTER(vp)]
        c.hz_component_major_index.argtypes = [vp, vp, u64]
        c.hz_component_major_index.restype = u64
        c.hz_witness_export_dev.argtypes = [vp, vp, ctypes.c_int32, vp, vp]
        c.hz_witness_export_range_dev.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, vp, vp]
        c.hz_witness_export_host.argtypes = [vp, vp, ctypes.c_int32, u64, u64, vp]
        c.hz_symmap_dev_index.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(u64)]
        c.hz_witness_derive_dev.argtypes = [vp, vp, ctypes.c_int32, ctypes.POINTER(vp), vp]
        c.hz_symbol_lookup.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(u64)]
        c.hz_constraint_name.restype = ctypes.c_char_p
        c.hz_ctx_set_shard.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        c.hz_da_record_bytes.argtypes = [vp]
        c.hz_da_record_bytes.restype = u64
        c.hz_da_export.argtypes = [vp, vp, vp]
        c.hz_da_import.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, vp]
        c.hz_witness_enqueue_tail.argtypes = [vp, vp]
        c.hz_witness_enqueue_tail_chain.argtypes = [vp, vp]
        for f in ("hz_sha_blocks", "hz_sha_state_bytes"):
            getattr(c, f).argtypes = [vp]
            getattr(c, f).restype = u64
        c.hz_sha_export.argtypes = [vp, vp, vp]
        c.hz_sha_expand.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, vp]
        c.hz_comm_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, ctypes.POINTER(vp)]
        c.hz_comm_destroy.argtypes = [vp]
        c.hz_comm_destroy.restype = None
        c.hz_comm_rank.argtypes = [vp]
        c.hz_comm_world.argtypes = [vp]
        c.hz_shard_step.argtypes = [vp, vp, vp]
        c.hz_ctx_ntx.argtypes = [vp]
        c.hz_ctx_set_profiling.argtypes = [vp, ctypes.c_int32]
        c.hz_profile_count.argtypes = [vp]
        c.hz_profile_get.argtypes = [vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(u64), ctypes.POINTER(u64)]

    def _check(self, st):
        if st != 0:
            raise HzError(st, self.c.hz_last_error().decode())

    def version(self):
        return self.c.hz_version().decode()

    def device_count(self):
        return self.c.hz_device_count()

    def poseidon_batch(self, t, inputs, witness=False, device=0):
        """inputs: list of n lists of t-1 ints -> (digests, sbox_witness or None)."""
        n = len(inputs)
        flat = fr_to_bytes([x for row in inputs for x in row])
        out = ctypes.create_string_buffer(32 * max(n, 1))
        nsbox = 8 * t + [56, 57, 56, 60, 60, 63][t - 2]
        wit = ctypes.create_string_buffer(96 * nsbox * max(n, 1)) if witness else None
        self._check(self.c.hz_poseidon_batch(device, t, n, flat, out, wit))
        return fr_from_bytes(out.raw[:32 * n]), (wit.raw if witness else None)

    def poseidon_batch_bytes(self, t, n, in_bytes, device=0):
        """n permutations from a bytes-like of n * (t-1) canonical 32-byte elements -> bytes of n digests (no per-element Python)"""
        out = ctypes.create_string_buffer(32 * max(n, 1))
        buf = (ctypes.c_char * len(in_bytes)).from_buffer_copy(in_bytes) if not isinstance(in_bytes, bytes) else in_bytes
        self._check(self.c.hz_poseidon_batch(device, t, n, buf, out, None))
        return out.raw[:32 * n]

    def poseidon_batch_dev(self, t, n, d_in, d_out, d_wit=None, stream=None):
        self._check(self.c.hz_poseidon_batch_dev(t, n, d_in, d_out, d_wit, stream))

    def template_device_bytes(self, template, nTx=0, nLevels=0, maxL1Tx=0, maxFeeTx=0, n_instances=1):
        """device memory a context of this shape will hold (hz_template_device_bytes): no device needed"""
        p = hz_params(TEMPLATES[template], nTx, nLevels, maxL1Tx, maxFeeTx, 0, n_instances, 0)
        self.c.hz_template_device_bytes.restype = ctypes.c_uint64
        self.c.hz_template_device_bytes.argtypes = [ctypes.POINTER(hz_params)]
        return self.c.hz_template_device_bytes(ctypes.byref(p))

    def fr_ops(self, op, a, b=None, device=0):
        """out[i] = a[i] (op) b[i]; op: 0 add, 1 sub, 2 mul, 3 sqr, 4 inv, 5 a*b+a+b, 6 2a*(-b)."""
        n = len(a)
        out = ctypes.create_string_buffer(32 * max(n, 1))
        self._check(self.c.hz_fr_ops(device, op, n, fr_to_bytes(a), fr_to_bytes(b) if b is not None else None, out))
        return fr_from_bytes(out.raw[:32 * n])

    def poseidon_dag(self, vals, job_in, job_out, seg_t, seg_first, seg_count, device=0):
        """vals: bytearray of 32-byte elements (known values in, digests out); job_in uint32[n,6], job_out uint32[n];
        segments (seg_t uint32, seg_first / seg_count uint64) in execution order. Returns the device time in ms."""
        import numpy as np
        assert job_in.dtype == np.uint32 and job_out.dtype == np.uint32 and seg_t.dtype == np.uint32
        assert seg_first.dtype == np.uint64 and seg_count.dtype == np.uint64 and job_in.flags.c_contiguous
        buf = (ctypes.c_char * len(vals)).from_buffer(vals)
        ms = ctypes.c_double(0.0)
        self._check(self.c.hz_poseidon_dag(device, buf, len(vals) // 32, job_in.ctypes.data_as(ctypes.c_void_p), job_out.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.c_uint64(len(job_out)), seg_t.ctypes.data_as(ctypes.c_void_p), seg_first.ctypes.data_as(ctypes.c_void_p),
                                           seg_count.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(seg_t)), ctypes.byref(ms)))
        return ms.value

    def ctx(self, template, **kw):
        return Ctx(self, template, **kw)

    def host_alloc(self, nbytes):
        """pinned host memory for hz_inputs_upload (address as int); free with host_free"""
        p = self.c.hz_host_alloc(nbytes)
        if not p:
            raise HzError(2, self.c.hz_last_error().decode())
        return p

    def host_free(self, p):
        self.c.hz_host_free(p)

    def shard_range(self, n_tx, world, rank):
        f, c = ctypes.c_int32(), ctypes.c_int32()
        self.c.hz_shard_range(n_tx, world, rank, ctypes.byref(f), ctypes.byref(c))
        return f.value, c.value


class Comm:
    """hz_comm: one per rank. transport "rccl" (librccl.so loaded by the library with dlopen) or "socket" (host-staged over the rendezvous)"""

    def __init__(self, L, transport, rank, world, path=None, device=0):
        self.L = L
        self.h = ctypes.c_void_p()
        L._check(L.c.hz_comm_create({"rccl": 1, "socket": 2}[transport], device, rank, world, path.encode() if path else None, ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.L.c.hz_comm_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_inputs(layout, inputs):
    """The packed bulk-upload buffer (bytes) of one instance from an input object {signal: number | nested lists} and
    Ctx.packed_layout(). Pure Python: usable in worker processes that have no GPU."""
    total, sigs = layout
    out = bytearray(total)
    for name, off, width, flat_len in sigs:
        flat = _flatten(inputs[name])
        if len(flat) != flat_len:
            raise ValueError("input %s: expected %d values, got %d" % (name, flat_len, len(flat)))
        if width == 32:
            out[off:off + 32 * flat_len] = fr_to_bytes(flat)
        else:
            out[off:off + flat_len] = bytes(flat)
    return bytes(out)


def _flatten(v):
    if isinstance(v, (list, tuple)):
        out = []
        for x in v:
            out.extend(_flatten(x))
        return out
    return [int(v)]


class Ctx:
    """One circuit context (hz_ctx): the object the reference's `tester()` returns."""

    def __init__(self, L, template, nTx=0, nLevels=0, maxL1Tx=0, maxFeeTx=0, n_instances=1, device=0, flags=0):
        self.L = L
        self.h = ctypes.c_void_p()
        p = hz_params(TEMPLATES[template], nTx, nLevels, maxL1Tx, maxFeeTx, device, n_instances, flags)
        L._check(L.c.hz_ctx_create(ctypes.byref(p), ctypes.byref(self.h)))
        self.n_instances = n_instances

    def close(self):
        if self.h:
            self.L.c.hz_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def witness_len(self):
        return self.L.c.hz_witness_len(self.h)

    def total(self):
        return self.L.c.hz_witness_total(self.h)

    def device_bytes(self):
        """device memory of this context, buffers allocated on first use included (hz_ctx_device_bytes)"""
        return self.L.c.hz_ctx_device_bytes(self.h)

    def constraint_estimate(self):
        return self.L.c.hz_constraint_estimate(self.h)

    def input_names(self):
        n = self.L.c.hz_input_count(self.h)
        out = []
        for i in range(n):
            ln = ctypes.c_uint64()
            out.append((self.L.c.hz_input_name(self.h, i, ctypes.byref(ln)).decode(), ln.value))
        return out

    def set_input(self, name, value, instance=0):
        flat = _flatten(value)
        self.L._check(self.L.c.hz_set_input(self.h, instance, name.encode(), fr_to_bytes(flat), len(flat)))

    def set_inputs(self, d, instance=0):
        for k, v in d.items():
            self.set_input(k, v, instance)

    def packed_layout(self):
        """(total bytes, [(name, byte offset, element bytes, flat length)]) of the bulk-upload buffer of one instance"""
        total = self.L.c.hz_inputs_packed_bytes(self.h)
        if not total:
            raise HzError(2, self.L.c.hz_last_error().decode())
        return total, [(nm, self.L.c.hz_input_packed_offset(self.h, i), self.L.c.hz_input_packed_width(self.h, i), ln) for i, (nm, ln) in enumerate(self.input_names())]

    def upload(self, instance, packed, nbytes=None, stream=None):
        """hz_inputs_upload: `packed` is bytes / bytearray, or the address of a (pinned) host buffer of nbytes"""
        if isinstance(packed, (bytes, bytearray)):
            nbytes = len(packed)
            buf = (ctypes.c_char * nbytes).from_buffer_copy(packed)
            self.L._check(self.L.c.hz_inputs_upload(self.h, instance, ctypes.addressof(buf), nbytes, stream))
            self._keep = getattr(self, "_keep", [])[-63:] + [buf]   # pageable source: keep it alive until the copy has run
        else:
            self.L._check(self.L.c.hz_inputs_upload(self.h, instance, packed, nbytes, stream))

    def stage(self, instance, packed_addr, nbytes, stream=None):
        """hz_inputs_stage: the H2D copy alone (pinned source); the next enqueue scatters the staged instances"""
        self.L._check(self.L.c.hz_inputs_stage(self.h, instance, packed_addr, nbytes, stream))

    def stage_range(self, first, count, packed_addr, nbytes_each, stride=None, stream=None):
        """hz_inputs_stage_range: `count` consecutive instances, one copy when the host buffers are contiguous"""
        self.L._check(self.L.c.hz_inputs_stage_range(self.h, first, count, packed_addr, nbytes_each, nbytes_each if stride is None else stride, stream))

    def copy_instance_inputs(self, src, dst, stream=None):
        """Replicate the inputs of instance `src` onto instance `dst` on the device."""
        self.L._check(self.L.c.hz_copy_instance_inputs(self.h, src, dst, stream))

    def clear_inputs(self):
        self.L.c.hz_clear_inputs(self.h)

    def _raise(self, st, err):
        if st == 3:
            raise ConstraintError(err.instance, err.unit, err.constraint_id, self.L.c.hz_constraint_name(err.constraint_id).decode(),
                                  int.from_bytes(bytes(err.lhs), "little"), int.from_bytes(bytes(err.rhs), "little"))
        self.L._check(st)

    def run(self):
        err = hz_error()
        self._raise(self.L.c.hz_witness_run(self.h, ctypes.byref(err)), err)

    def enqueue(self, stream=None):
        self.L._check(self.L.c.hz_witness_enqueue(self.h, stream))

    def check(self):
        err = hz_error()
        self._raise(self.L.c.hz_witness_check(self.h, ctypes.byref(err)), err)

    def failures(self):
        """After run() / check(): the first violated constraint of every failing instance, ordered by instance, as
        (instance, unit, constraint_id, name, lhs, rhs) tuples (hz_witness_failures)."""
        n = ctypes.c_size_t(0)
        self.L._check(self.L.c.hz_witness_failures(self.h, None, 0, ctypes.byref(n)))
        if n.value == 0:
            return []
        arr = (hz_error * n.value)()
        self.L._check(self.L.c.hz_witness_failures(self.h, arr, n.value, ctypes.byref(n)))
        return [(e.instance, e.unit, e.constraint_id, self.L.c.hz_constraint_name(e.constraint_id).decode(),
                 int.from_bytes(bytes(e.lhs), "little"), int.from_bytes(bytes(e.rhs), "little")) for e in arr]

    def set_inputs_json(self, text, instance=0):
        b = text.encode() if isinstance(text, str) else text
        self.L._check(self.L.c.hz_set_inputs_json(self.h, instance, b, len(b)))

    def write_wtns(self, path, instance=0):
        self.L._check(self.L.c.hz_witness_write_wtns(self.h, instance, path.encode()))

    def write_json(self, path, instance=0):
        self.L._check(self.L.c.hz_witness_write_json(self.h, instance, path.encode()))

    def write_sym(self, path):
        self.L._check(self.L.c.hz_symbols_write_sym(self.h, path.encode()))

    def read(self, first, count, instance=0):
        buf = ctypes.create_string_buffer(32 * max(count, 1))
        self.L._check(self.L.c.hz_witness_read(self.h, instance, first, count, buf))
        return fr_from_bytes(buf.raw[:32 * count])

    def read_bytes(self, first, count, instance=0):
        """the same elements as read(), as 32-byte little-endian records (no Python integers: GB-sized compares)"""
        buf = ctypes.create_string_buffer(32 * max(count, 1))
        self.L._check(self.L.c.hz_witness_read(self.h, instance, first, count, buf))
        return buf.raw[:32 * count]

    def read_raw_bytes(self, first=0, count=None):
        count = self.total() - first if count is None else count
        buf = ctypes.create_string_buffer(32 * max(count, 1))
        self.L._check(self.L.c.hz_witness_read_raw(self.h, first, count, buf))
        return buf.raw[:32 * count]

    # -- multi-GPU intra-batch shard
    def set_shard(self, first, count, tail):
        self.L._check(self.L.c.hz_ctx_set_shard(self.h, first, count, 1 if tail else 0))

    def da_record_bytes(self):
        return self.L.c.hz_da_record_bytes(self.h)

    def da_export(self, d_buf, stream=None):
        self.L._check(self.L.c.hz_da_export(self.h, d_buf, stream))

    def da_import(self, first, count, d_buf, stream=None):
        self.L._check(self.L.c.hz_da_import(self.h, first, count, d_buf, stream))

    def enqueue_tail(self, stream=None):
        self.L._check(self.L.c.hz_witness_enqueue_tail(self.h, stream))

    def enqueue_tail_chain(self, stream=None):
        self.L._check(self.L.c.hz_witness_enqueue_tail_chain(self.h, stream))

    def sha_blocks(self):
        return self.L.c.hz_sha_blocks(self.h)

    def sha_state_bytes(self):
        return self.L.c.hz_sha_state_bytes(self.h)

    def sha_export(self, d_buf, stream=None):
        self.L._check(self.L.c.hz_sha_export(self.h, d_buf, stream))

    def sha_expand(self, first, count, d_buf=None, stream=None):
        self.L._check(self.L.c.hz_sha_expand(self.h, first, count, d_buf, stream))

    def shard_step(self, comm, stream):
        """hz_shard_step: one sharded pass on `stream` (a hipStream_t handle) through the communicator `comm` (Comm); then check()"""
        self.L._check(self.L.c.hz_shard_step(self.h, comm.h, stream))

    def set_profiling(self, on=True, exclusive=False):
        self.L._check(self.L.c.hz_ctx_set_profiling(self.h, (2 if exclusive else 1) if on else 0))

    def profile(self):
        """[(kernel, ms, algorithmic_bytes, units)] of the last enqueue (after check())."""
        out = []
        for i in range(self.L.c.hz_profile_count(self.h)):
            nm, ms, by, un = ctypes.c_char_p(), ctypes.c_float(), ctypes.c_uint64(), ctypes.c_uint64()
            self.L._check(self.L.c.hz_profile_get(self.h, i, ctypes.byref(nm), ctypes.byref(ms), ctypes.byref(by), ctypes.byref(un)))
            out.append((nm.value.decode(), ms.value, by.value, un.value))
        return out

    def dev_ptr(self):
        return self.L.c.hz_witness_dev_ptr(self.h)

    def lookup(self, name):
        idx = ctypes.c_uint64()
        if not self.L.c.hz_symbol_lookup(self.h, name.encode(), ctypes.byref(idx)):
            raise KeyError(name)
        return idx.value

    def get(self, name, instance=0):
        return self.read(self.lookup(name), 1, instance)[0]

    def import_sym(self, text, r1cs=None):
        """circom .sym text (and, optionally, the .r1cs bytes of the same compile: hz_symmap_create_r1cs) -> SymMap (the witness in the
        compiler's variable order)"""
        b = text.encode() if isinstance(text, str) else text
        h = ctypes.c_void_p()
        if r1cs is None:
            self.L._check(self.L.c.hz_symmap_create(self.h, b, len(b), ctypes.byref(h)))
        else:
            self.L._check(self.L.c.hz_symmap_create_r1cs(self.h, b, len(b), bytes(r1cs), len(r1cs), ctypes.byref(h)))
        return SymMap(self, h)

    def symmap_from_index(self, index):
        """hz_symmap_from_index: `index` a numpy uint64 array, variable v = stored signal index[v] (index[0] == 0)"""
        import numpy as np
        a = np.ascontiguousarray(index, dtype=np.uint64)
        h = ctypes.c_void_p()
        self.L._check(self.L.c.hz_symmap_from_index(self.h, a.ctypes.data, a.size, ctypes.byref(h)))
        return SymMap(self, h)

    def component_major_index(self):
        """hz_component_major_index -> numpy uint64 array"""
        import numpy as np
        n = self.L.c.hz_component_major_index(self.h, None, 0)
        a = np.empty(n, dtype=np.uint64)
        self.L.c.hz_component_major_index(self.h, a.ctypes.data, n)
        return a

    def load_symmap(self, path):
        """hz_symmap_load: a map written by SymMap.save for this template and shape"""
        h = ctypes.c_void_p()
        self.L._check(self.L.c.hz_symmap_load(self.h, path.encode(), ctypes.byref(h)))
        return SymMap(self, h)

    def symbol_count(self):
        return self.L.c.hz_symbol_count(self.h)

    def symbol(self, i):
        s = hz_symbol()
        self.L._check(self.L.c.hz_symbol_get(self.h, i, ctypes.byref(s)))
        return s.name.decode(), s.index


class SymMap:
    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def __del__(self):
        try:
            self.ctx.L.c.hz_symmap_destroy(self.h)
        except Exception:
            pass

    def nvars(self):
        return self.ctx.L.c.hz_symmap_nvars(self.h)

    def unresolved(self):
        """[(variable, a label)] of the variables none of whose labels the layout stores"""
        out = []
        n = self.ctx.L.c.hz_symmap_unresolved(self.h, 0, None, None)
        for i in range(n):
            v, nm = ctypes.c_uint64(), ctypes.c_char_p()
            self.ctx.L.c.hz_symmap_unresolved(self.h, i, ctypes.byref(v), ctypes.byref(nm))
            out.append((v.value, nm.value.decode()))
        return out

    def save(self, path):
        self.ctx.L._check(self.ctx.L.c.hz_symmap_save(self.ctx.h, self.h, path.encode()))

    def solved(self):
        """variables defined by the linear constraints of the .r1cs (hz_symmap_create_r1cs)"""
        return self.ctx.L.c.hz_symmap_solved(self.h)

    def check_r1cs(self, instance=0, cap=16):
        """(number of violated constraints, indices of the first `cap`) of the map's .r1cs on the witness it serves"""
        n, first = ctypes.c_uint64(), (ctypes.c_uint64 * cap)()
        self.ctx.L._check(self.ctx.L.c.hz_symmap_check_r1cs(self.ctx.h, self.h, instance, ctypes.byref(n), first, cap))
        return n.value, list(first[:min(cap, n.value)])

    def derived(self):
        """variables evaluated from stored signals by a rule (linear signals an unreduced compile keeps)"""
        f = self.ctx.L.c.hz_symmap_derived
        f.restype, f.argtypes = ctypes.c_uint64, [ctypes.c_void_p]
        return f(self.h)

    def read(self, first=0, count=None, instance=0):
        count = self.nvars() - first if count is None else count
        buf = ctypes.create_string_buffer(32 * max(count, 1))
        self.ctx.L._check(self.ctx.L.c.hz_witness_read_sym(self.ctx.h, self.h, instance, first, count, buf))
        return fr_from_bytes(buf.raw[:32 * count])

    def write_wtns(self, path, instance=0):
        self.ctx.L._check(self.ctx.L.c.hz_witness_write_wtns_sym(self.ctx.h, self.h, instance, path.encode()))

    def read_small(self, first, count, instance=0):
        """hz_witness_read_sym in pieces small enough to stay on its host evaluator (the independent route the device export is
        compared with in tests)"""
        out = []
        for f in range(first, first + count, 4096):
            out += self.read(f, min(4096, first + count - f), instance)
        return out

    def upload(self):
        """hz_symmap_upload: the map's device tables for this context; returns their size in bytes"""
        n = ctypes.c_uint64()
        self.ctx.L._check(self.ctx.L.c.hz_symmap_upload(self.ctx.h, self.h, ctypes.byref(n)))
        return n.value

    def export_dev(self, d_out, instance=0, stream=None):
        """hz_witness_export_dev: the witness of `instance` (-1: every instance) in this map's variable order into device memory"""
        self.ctx.L._check(self.ctx.L.c.hz_witness_export_dev(self.ctx.h, self.h, instance, d_out, stream))

    def export_host(self, instance=0, first=0, count=None, out=None):
        """hz_witness_export_host -> bytes (or into the caller's buffer address `out`)"""
        count = self.nvars() - first if count is None else count
        if out is not None:
            self.ctx.L._check(self.ctx.L.c.hz_witness_export_host(self.ctx.h, self.h, instance, first, count, out))
            return None
        buf = ctypes.create_string_buffer(32 * max(count, 1))
        self.ctx.L._check(self.ctx.L.c.hz_witness_export_host(self.ctx.h, self.h, instance, first, count, buf))
        return buf.raw[:32 * count]

    def dev_index(self):
        """hz_symmap_dev_index -> (device pointer to u64 phys0[nvars], device pointer to u32 inst_stride[nvars], derived slots)"""
        a, b, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        self.ctx.L._check(self.ctx.L.c.hz_symmap_dev_index(self.ctx.h, self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
        return a.value, b.value, n.value

    def derive_dev(self, instance=0, stream=None):
        p = ctypes.c_void_p()
        self.ctx.L._check(self.ctx.L.c.hz_witness_derive_dev(self.ctx.h, self.h, instance, ctypes.byref(p), stream))
        return p.value


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = Lib()
    return _lib
