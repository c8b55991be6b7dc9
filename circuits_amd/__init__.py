"""circuits_amd -- MI355X-native witness generator for the Hermez rollup circuits.

The compute path is hand-written HIP for gfx950 in ``circuits_amd/csrc`` behind the C ABI of
``include/hermez_witness.h`` (``libhermez_witness.so``). This package is only the thin Python
binding used by the tests and the benchmark; the Node.js facade in ``circuits_amd/node`` mirrors
the reference's ``tester()/calculateWitness()/assertOut()`` surface over the same ABI.
There is no CPU fallback: importing works everywhere, computing needs a gfx950 device.
"""
from .capi import (  # noqa: F401
    HzError,
    ConstraintError,
    Lib,
    Ctx,
    lib,
    lib_path,
    fr_to_bytes,
    fr_from_bytes,
    R_MODULUS,
)
