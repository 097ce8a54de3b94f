"""Test-only launcher: biscuit_amd.multi_gpu's sharding + streaming gather over gloo, with the aligner entry
point replaced by the CPU restatement under oracle/ (there is no GPU in the CPU test container).  The
product package itself never opens anything under oracle/."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from biscuit_amd import multi_gpu, _lib
    _lib.lib()   # libbiscuit_amd.so first (RTLD_GLOBAL): the restatement links against its host code
    entry = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_port.so")).oracle_align_main
    sys.exit(multi_gpu.main(sys.argv[1:], entry=entry, use_gpu=False))
