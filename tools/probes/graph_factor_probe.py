"""diagnostic: ONE factor object factors several different matrices in a row (first run eager, second captured, later ones replayed from the
hipGraph) -- every factor against numpy; with and without HDM_POISON, with and without HDM_GRAPHS, the host matrix given with a clean
or a NaN upper triangle (the engine's dual matrices are lower-valid)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np
    from hdsdp_amd import api
    n = int(sys.argv[1]); nan_upper = sys.argv[2] == "nan"
    rng = np.random.default_rng(3)
    ls = api.LinSys(n)
    out = []
    for k in range(5):
        G = rng.standard_normal((n, n))
        S = G @ G.T + n * np.eye(n) * (1 + k)
        if k in (1, 2) and os.environ.get("PROBE_BAD"):
            S[n // 3, n // 3] = -5.0                    # not positive definite: the captured / first replayed run fails half way
        buf = np.triu(S).copy()                      # C-order upper == column-major lower
        if nan_upper:
            buf[np.tril_indices(n, -1)] = np.nan
        ok = ls.psd_check(buf)
        if not ok:
            out.append((False,)); continue
        d = ls.get_diag()
        ref = np.diag(np.linalg.cholesky(S))
        x = ls.solve(np.ones((1, n)))
        out.append((bool(ok), float(np.max(np.abs(d - ref)) / np.max(ref)), float(np.max(np.abs(S @ x.T - 1.0)))))
    print(n, sys.argv[2], {k: os.environ.get(k) for k in ("HDM_POISON", "HDM_GRAPHS")}, out, flush=True)
else:
    for n in (2000,):
        for up in ("nan",):
            for extra in ({"PROBE_BAD": "1"}, {"PROBE_BAD": "1", "HDM_POISON": "1"}, {"PROBE_BAD": "1", "HDM_POISON": "1", "HDM_GRAPHS": "0"}):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), up], env=dict(os.environ, **extra), capture_output=True, text=True)
                lines = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(str(n))]
                print(lines[-1][:400] if lines else (r.stdout + r.stderr)[-400:], flush=True)
