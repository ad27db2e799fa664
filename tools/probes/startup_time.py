"""where the first 0.25 s of a process go: library load, device context, first launch, first cone"""
import os, sys, time
t0 = time.perf_counter()
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
t1 = time.perf_counter()
lib = C.CDLL(os.path.join(ROOT, "hdsdp_amd", "libhdsdp_mi355x.so"))
t2 = time.perf_counter()
lib.HMiDeviceInit.restype = C.c_int
lib.HMiDeviceInit(0)
t3 = time.perf_counter()
lib.HMiDiagBlockProbe.restype = C.c_double
lib.HMiDiagBlockProbe.argtypes = [C.c_int, C.c_int]
lib.HMiDiagBlockProbe(1, 1)
t4 = time.perf_counter()
lib.HMiDiagBlockProbe(1, 1)
t5 = time.perf_counter()
print(f"imports {1e3*(t1-t0):.0f} ms | dlopen (librccl, libamdhip64 come with it) {1e3*(t2-t1):.0f} ms | HMiDeviceInit (runtime, context, stream, events) {1e3*(t3-t2):.0f} ms | "
      f"first kernel launches (code object load) {1e3*(t4-t3):.0f} ms | the same again {1e3*(t5-t4):.1f} ms")
