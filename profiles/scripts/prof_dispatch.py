"""Workgroup dispatch rate of the device (mcba_debug_dispatch_probe): how long the GPU takes to START n workgroups."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, ".")
from multical_amd import _lib
from multical_amd.backend import check
lib = _lib.load()
print("blocks threads lds_KB spin | start spread us | ns per block | kernel span us (first start .. last end)")
for blocks, threads, lds, spin in [(512, 64, 0, 0), (1024, 64, 0, 0), (4096, 64, 0, 0), (4096, 64, 16, 0), (512, 256, 0, 0), (1024, 256, 0, 0),
                                   (1024, 256, 44, 0), (512, 256, 44, 0), (256, 1024, 0, 0), (128, 1024, 48, 0), (4096, 64, 16, 2000),
                                   (2048, 64, 16, 2000), (1024, 256, 0, 2000), (1024, 256, 44, 2000)]:
  out = np.zeros((blocks, 2), dtype=np.int64)
  check(lib.mcba_debug_dispatch_probe(blocks, threads, lds * 1024, spin, out.ctypes.data_as(C.POINTER(C.c_longlong))))
  st = out[:, 0] - out[:, 0].min()
  print(f"{blocks:6d} {threads:7d} {lds:6d} {spin:5d} | {st.max() / 100:8.2f} | {st.max() * 10 / blocks:8.2f} | {(out[:, 1].max() - out[:, 0].min()) / 100:8.2f}")
