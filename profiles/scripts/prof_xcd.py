"""Is data written by workgroup w of a kernel still in the L2 of w's XCD for the next kernel?  (mcba_debug_xcd_probe)"""
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multical_amd import _lib
lib = _lib.load()
for n in (2048, 16384, 65536):
    out = (C.c_longlong * 16)()
    _lib.check(lib.mcba_debug_xcd_probe(n, out))
    print("n = %6d doubles (%4d KB): cycles to read the region written by workgroup 0..7: %s | second read of region 7: %d"
          % (n, n * 8 // 1024, " ".join(str(out[i]) for i in range(8)), out[8]), flush=True)
