import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hdsdp_amd import api
lib = api.load_library()
for rep in range(2):
    print("constant operands, 16 acc GEMM pattern, 2 waves/SIMD: %.1f TF" % lib.HMiMfmaIssueProbe(0, 2, 40000))
    print("pseudo-random operands,      same loop, 2 waves/SIMD: %.1f TF" % lib.HMiMfmaIssueProbe(300, 2, 40000))
