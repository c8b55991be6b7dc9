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
        so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
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
