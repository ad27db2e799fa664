"""diagnostic (HDM_VAR=96: stamps on the default loop): per-workgroup timeline of the last launch of one role"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HDM_VAR", "96")
import torch
from hdsdp_amd import api
role = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = m = 2000
lib = api.load_library()
cone = api.SDPCone.synthetic(n, m)
kkt = api.KKT(m, [cone], host_mirror=False)
cone.set_start(-10.0 * n)
assert cone.check_is_interior(1.0, np.zeros(m))
kkt.build_up(0)
nwg = 1 << 18
dbg = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
lib.HMiSetDebugBuffer(dbg.data_ptr(), role)
kkt.build_up(0)   # every launch of that role overwrites the buffer: the last one stays
lib.HMiSetDebugBuffer(None, -1)
lib.HMiDeviceSynchronize()
d = dbg.cpu().numpy().reshape(nwg, 8)
d = d[d[:, 0] != 0]
print("workgroups stamped:", len(d))
from wg_timeline_gemm import residency
residency(d, "role %d in the pipeline" % role)
t0 = d[:, 0].min()
start, pro, loop, end = (d[:, i] - t0 for i in range(4))
hw, xcc, nst = d[:, 4], d[:, 5], d[:, 6]
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | ((xcc & 0xF) << 8)
real = d[:, 7]
for x in np.unique(xcc)[:2]:
    sx = xcc == x
    ia, ib = np.argmin(real[sx]), np.argmax(real[sx])
    dt_real = (real[sx][ib] - real[sx][ia]) / 100e6
    print("xcc %d: memtime ticks per second = %.4e ; span %.3f ms" % (x, (d[sx][ib, 3] - d[sx][ia, 3]) / dt_real, dt_real * 1e3))
tot = end.max()
mt = d[:, 6].astype(np.int64) >= 0
print("main tiles only:")
start, pro, loop, end, nst = start[mt], pro[mt], loop[mt], end[mt], nst[mt]
print("mean per WG: prologue %.0f  loop %.0f  epilogue %.0f ticks ; loop ticks per stage %.1f" % (
    (pro - start).mean(), (loop - pro).mean(), (end - loop).mean(), ((loop - pro) / np.maximum(nst, 1)).mean()))
# per-CU occupancy timeline: fraction of the span during which 0 / 1 / 2 WGs are in their loop
ucu = np.unique(cu)
print("distinct CU ids:", len(ucu))
frac = np.zeros(3)
for c in ucu:
    sel = cu == c
    base = d[sel][:, 0].min()
    p_, l_ = d[sel][:, 1] - base, d[sel][:, 2] - base
    ev = sorted([(t, +1) for t in p_] + [(t, -1) for t in l_])
    cur, last = 0, 0
    for t, s_ in ev:
        frac[min(cur, 2)] += t - last
        cur += s_
        last = t
frac /= frac.sum()
print("CU time with 0 / 1 / >=2 workgroups inside the MFMA loop: %.3f %.3f %.3f" % tuple(frac))
# residency (start .. end of every workgroup, cell-dealt tiles included): how much of a CU's time has fewer than two
# workgroups on it -- the dispatcher's hand-over gap that persistent workgroups would remove
res = np.zeros(4)
for c in ucu:
    sel = cu == c
    base = d[sel][:, 0].min()
    s_, e_ = d[sel][:, 0] - base, d[sel][:, 3] - base
    ev = sorted([(t, +1) for t in s_] + [(t, -1) for t in e_])
    cur, last = 0, ev[0][0]
    for t, k_ in ev:
        res[min(cur, 3)] += t - last
        cur += k_
        last = t
res /= res.sum()
kinds = d[:, 6].astype(np.int64)
print("workgroups by kind: main %d, diagonal %d, edge %d" % ((kinds >= 0).sum(), (kinds == -1).sum(), (kinds == -2).sum()))
print("CU time with 0 / 1 / 2 / >2 workgroups RESIDENT: %.4f %.4f %.4f %.4f" % tuple(res))
print("first WG start spread: %d ticks; last end: %d" % (start.min(), end.max()))
# loop duration per stage as a function of co-residency is in the raw file
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "wg_timeline_role%d.npy" % role), d)
ps = (loop - pro) / np.maximum(nst, 1)
print("per-stage loop ticks percentiles 5/25/50/75/95: " + " ".join("%.0f" % v for v in np.percentile(ps, [5, 25, 50, 75, 95])))
for lo, hi in ((1, 16), (17, 48), (49, 96), (97, 200)):
    sel = (nst >= lo) & (nst <= hi)
    if sel.any():
        print("  tiles with %3d-%3d stages: n=%4d per-stage %.0f  epilogue %.0f  prologue %.0f" % (
            lo, hi, sel.sum(), ps[sel].mean(), (end - loop)[sel].mean(), (pro - start)[sel].mean()))
# dispatch gap: for every workgroup end, the delay until the next workgroup start on the same CU
gaps = []
for c in ucu:
    sel = cu == c
    st = np.sort(d[sel][:, 0]); en = np.sort(d[sel][:, 3])
    j = np.searchsorted(st, en, side="left")
    ok = j < len(st)
    gaps.extend((st[j[ok]] - en[ok]).tolist())
gaps = np.array(gaps)
print("end -> next start on the same CU (ticks): median %.0f mean %.0f p90 %.0f  (n=%d)" % (
    np.median(gaps), gaps.mean(), np.percentile(gaps, 90), len(gaps)))
c = ucu[3]
sel = np.where(cu == c)[0]
base = d[sel][:, 0].min()
order = sel[np.argsort(d[sel][:, 0])]
print("timeline of CU %d (ticks since first start): start pro loop end nst" % c)
for i in order[:14]:
    print("   ", *(int(d[i, k] - base) for k in range(4)), int(d[i, 6]))
