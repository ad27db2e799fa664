"""fp64 MFMA issue ceiling by occupancy: register-only loops shaped like the GEMM inner loop (HMiMfmaIssueProbe)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hdsdp_amd import api
lib = api.load_library()
lib.HMiDeviceInit(0)
print("peak probe (8 acc, 8 WG/CU):", round(lib.HMiMfmaPeakProbe(20000), 2), "TFLOP/s")
for mode, name in ((0, "16 acc, 4+4 operand regs (GEMM pattern)"), (1, "16 acc, one operand pair"), (2, "8 acc, GEMM pattern"), (300, "16 acc, random operands")):
    for wg in (1, 2, 3, 4):
        v = lib.HMiMfmaIssueProbe(mode, wg, 4000)
        v = lib.HMiMfmaIssueProbe(mode, wg, 4000)
        print(f"mode {mode:3d} ({name}), {wg} WG/CU = {wg} wave(s)/SIMD: {v:6.2f} TFLOP/s = {v / 78.6:.3f} of 78.6")
