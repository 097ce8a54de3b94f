"""Loaders for the checkers under oracle/ (test infrastructure only)."""
import ctypes as C
import os
import numpy as np
from biscuit_amd import _lib as B
from biscuit_amd.api import Batches

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORT_PATH = os.path.join(ROOT, "oracle", "liboracle_port.so")
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libbiscuit_ref.so")


def port_lib():
    B.lib()
    L = C.CDLL(PORT_PATH)
    L.oracle_port_new.restype = C.c_void_p
    L.oracle_port_new.argtypes = [C.c_void_p, C.c_int]
    L.oracle_sa.restype = C.c_uint64
    return L


def ref_lib():
    """The real reference functions (built from /root/reference by oracle/Makefile); None if absent."""
    if not os.path.exists(REF_PATH):
        return None
    R = C.CDLL(REF_PATH)
    R.ref_bwt_load.restype = C.c_void_p
    R.ref_bwt_sa.restype = C.c_uint64
    R.ref_bwt_cal_sa.restype = C.c_uint64
    R.ref_hash_64.restype = C.c_uint64
    R.ref_hash_64.argtypes = [C.c_uint64]
    R.ref_bt_new.restype = C.c_void_p
    return R


class Port(Batches):
    """CPU restatement of the device kernels behind the same batch seams (oracle/port.c)."""

    def __init__(self, index, n_threads=1):
        self.L = port_lib()
        self.index = index
        self.h = C.c_void_p(self.L.oracle_port_new(index.h, n_threads))
        L = self.L
        fns = {"set_opt": L.oracle_port_set_opt, "set_reads": L.oracle_port_set_reads, "seed_batch": L.oracle_port_seed_batch,
               "sa_batch": L.oracle_port_sa_batch, "extend_batch": L.oracle_port_extend_batch, "sw_batch": L.oracle_port_sw_batch,
               "global_batch": L.oracle_port_global_batch}
        Batches.__init__(self, fns, self.h)

    def use_reference_kernels(self):
        """the same seams over the reference's own kernels (oracle/_ref: bwt.c, ksw.c); False when the library is absent"""
        if not os.path.exists(REF_PATH):
            return False
        self.L.oracle_port_use_reference_kernels.argtypes = [C.c_void_p, C.c_char_p]
        return self.L.oracle_port_use_reference_kernels(self.h, REF_PATH.encode()) == 0

    def backend(self):
        be = B.Backend()
        self.L.oracle_port_backend(self.h, C.byref(be))
        return be

    def counters(self, reset=False):
        c = (C.c_uint64 * 4)()
        self.L.oracle_port_counters(self.h, c, int(reset))
        return list(c)
