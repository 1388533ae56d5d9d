"""TEST INFRASTRUCTURE ONLY: Python handle on the CPU oracle (oracle/mfk_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It reuses the generic
ctypes wrapper of the product (`mustache_amd.lib.Binding`) with the oracle's own shared object and the `mfko_` prefix,
so the oracle and the HIP library are driven through identical signatures and their results compare field by field.
"""
import os
import subprocess
import ctypes as C

import numpy as np

from mustache_amd import lib as _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "_build", "libmfk_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "mfk_oracle.c")
    if force or not os.path.exists(SO_PATH) or os.path.getmtime(SO_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "_build/libmfk_oracle.so"] + (["-B"] if force else []))
    return SO_PATH


class Oracle(_lib.Binding):
    def __init__(self):
        # on the GPU box the prebuilt .so travels with the tree; rebuild only when the source is newer
        try:
            build()
        except Exception:
            if not os.path.exists(SO_PATH):
                raise
        super().__init__(SO_PATH, "mfko_", 0)
        self.so.mfko_take_closest_smaller.restype = C.c_int
        self.so.mfko_take_closest_larger.restype = C.c_int
        self.so.mfko_levenshtein.restype = C.c_int

    def reads(self, packed):
        """Describe host (numpy) arrays of a PackedReads to the oracle."""
        arrays = {k: np.ascontiguousarray(v) for k, v in packed.a.items()}
        return self.reads_struct(arrays, packed.n_reads, packed.max_span, _lib.host_header(packed))

    def take_closest(self, values, number, larger):
        a = np.ascontiguousarray(values, dtype=np.int32)
        out = C.c_int32(0)
        fn = self.so.mfko_take_closest_larger if larger else self.so.mfko_take_closest_smaller
        ok = fn(C.c_void_p(a.ctypes.data), C.c_int64(a.shape[0]), C.c_int32(number), C.byref(out))
        return out.value if ok else None

    def levenshtein(self, a: bytes, b: bytes):
        return self.so.mfko_levenshtein(C.c_char_p(a), C.c_int(len(a)), C.c_char_p(b), C.c_int(len(b)))

    def ir_report(self, seq: str):
        """All inverted repeats the oracle's einverted restatement reports, 1-based inclusive arms, report order."""
        out = np.zeros(32, dtype=np.int32)
        self.so.mfko_ir_report.restype = C.c_int
        n = self.so.mfko_ir_report(C.c_char_p(seq.encode()), C.c_int(len(seq)), C.c_void_p(out.ctypes.data))
        return [tuple(int(x) for x in out[4 * k:4 * k + 4]) for k in range(n)]
