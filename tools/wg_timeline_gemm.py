"""diagnostic (HDM_VAR=32): workgroup residency of ONE stand-alone GEMM launch (role 0) in global time"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HDM_VAR", "32")
import torch
from hdsdp_amd import api


def residency(d, label):
    d = d[d[:, 0] != 0].astype(np.float64)
    hw = d[:, 4].astype(int); xcc = d[:, 5].astype(int)
    key = xcc * 1000 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 0xF)
    c0 = key == key[0]
    rate = np.polyfit(d[c0, 7], d[c0, 3], 1)[0] if c0.sum() > 2 else 23.0
    start = np.zeros(len(d)); end = d[:, 7].copy()
    for k in np.unique(key):
        s = key == k
        off = np.median(d[s, 3] - rate * d[s, 7])
        start[s] = (d[s, 0] - off) / rate
    t0 = start.min(); start -= t0; end -= t0
    span = end.max()
    print("%s: %d workgroups, span %.3f ms, memtime/realtime %.2f" % (label, len(d), span / 1e5, rate))
    print("   time-averaged resident workgroups: %.1f" % ((end - start).sum() / span))
    o = np.sort(start)
    print("   started within 0.05/0.2/0.5/1/2 ms: %s" % [int((o < t * 1e5).sum()) for t in (0.05, 0.2, 0.5, 1, 2)])
    ts = np.linspace(0, span, 12)[1:-1]
    print("   resident at 10 sample times:", [int(((start <= t) & (end > t)).sum()) for t in ts])


if __name__ == "__main__":
    lib = api.load_library()
    lib.HMiDeviceInit(0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    A = torch.randn(K, N, dtype=torch.float64, device="cuda")   # M-major: element (i,k) at i + k*lda
    B = torch.randn(K, N, dtype=torch.float64, device="cuda")
    Cm = torch.zeros(N, N, dtype=torch.float64, device="cuda")
    nwg = 1 << 18
    dbg = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(2):
        dbg.zero_(); torch.cuda.synchronize()
        lib.HMiSetDebugBuffer(dbg.data_ptr(), 0)
        rc = lib.HMiGemmNT(A.data_ptr(), N, 0, B.data_ptr(), N, 0, Cm.data_ptr(), N, N, N, K, 1.0, 0.0, 0, 0)
        lib.HMiSetDebugBuffer(None, -1)
        assert rc == 0
        residency(dbg.cpu().numpy().reshape(nwg, 8), "stand-alone GEMM %d^2 x %d, rep %d" % (N, K, rep))
