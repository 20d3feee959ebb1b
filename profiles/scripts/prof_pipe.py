"""FP64 VALU vs FP64 MFMA: do two waves of one SIMD overlap them?  (mcba_debug_pipe_probe)"""
import sys, ctypes as C
sys.path.insert(0, ".")
from multical_amd import _lib
from multical_amd.backend import check
lib = _lib.load()
for iters in (2000, 8000):
    ms = (C.c_double * 3)()
    check(lib.mcba_debug_pipe_probe(iters, ms))
    print("iters %d: all waves FMA %.3f ms | all waves MFMA %.3f ms | one FMA wave + one MFMA wave per SIMD %.3f ms" % (iters, ms[0], ms[1], ms[2]))
    # per SIMD: mode 0 = 2 waves x iters x 64 FMA, mode 1 = 2 waves x iters x 4 MFMA, mode 2 = 1 + 1
    print("   if independent pipes: mode 2 = max(m0, m1) / 2 = %.3f ms;  if one shared FP64 pipe: (m0 + m1) / 2 = %.3f ms" % (max(ms[0], ms[1]) / 2, (ms[0] + ms[1]) / 2))
