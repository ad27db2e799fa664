"""Every environment switch the shipped library reads (the table in include/hdsdp_mi355x.h) in its NON-default position: the
same numbers must come out of every device path.  The library reads a switch once per process, so each setting runs
tests/switch_worker.py in a child process; the parent compares with the default run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

# switch -> other position(s).  (Kept in step with the header's table by test_abi_cpu.py::test_every_environment_switch_is_listed...)
SETTINGS = [
    {"HDSDP_MI355X_AFFINE_S": "0"}, {"HDSDP_MI355X_AFFINE_S": "2"},
    {"HDSDP_MI355X_SMALL_CHECK": "0"},
    {"HDSDP_MI355X_ZS": "0"}, {"HDSDP_MI355X_ZS": "2"},                 # sweeps from the dense storage everywhere / from the zero-suppressed copy everywhere
    {"HDSDP_MI355X_SPARSE_KKT": "0"}, {"HDSDP_MI355X_KKT_RCM": "0"}, {"HDSDP_MI355X_KKT_ENVELOPE": "0"}, {"HDSDP_MI355X_KKT_TILES": "0"},
    {"HDSDP_MI355X_FORCE_GEMM": "1"},
    {"HDM_LANCZOS_WHOLE": "0"}, {"HDM_LANCZOS_WHOLE": "0", "HDM_LANCZOS_FUSED": "0"},
    {"HDM_LANCZOS_WHOLE": "0", "HDM_LANCZOS_FUSED": "0", "HDM_LANCZOS_GROUP": "0"}, {"HDM_LANCZOS_BIG": "0"},
    {"HDM_LANCZOS_BIG": "0", "HDM_LANCZOS_GROUP": "0"},
    {"HDM_SYM_COMBINE_SKY": "0"}, {"HDM_SHARE_T_SLABS": "0"}, {"HDM_NSPLIT": "24"}, {"HDM_BC": "8"}, {"HDM_TCAP_GIB": "1"},
    {"HDM_GRAM_QUEUE": "0"},                                             # Gram jobs from one queue per XCD (the form up to round 4)
    {"HDM_GRAM_KSTAGES": "16", "HDM_NSPLIT": "8"},                       # more K splits than slabs: groups of 8 accumulate into the slabs
    {"HDSDP_MI355X_HOST_THREADS": "1"},                                  # ingest on one host thread
    {"HDM_PERSIST": "0"}, {"HDM_PERSIST_RESERVE_CUS": "200"}, {"HDSDP_MI355X_PRELOAD": "0"},
    {"HDM_DIAG_SWEEP": "0"}, {"HDM_CHOL_K128": "0"}, {"HDM_TRSV_FLOW": "0"}, {"HDM_GRAPHS": "1"}, {"HDM_GRAPHS": "2"},
    {"SWITCH_WORKER_SHARDS": "2"},                                       # in-process device group, defaults
    {"SWITCH_WORKER_SHARDS": "3", "HDSDP_MI355X_A2A_PIECES": "1"},       # one blocking exchange
    {"SWITCH_WORKER_SHARDS": "2", "HDSDP_MI355X_STAGED_A2A": "0"},       # congruence complete, then exchange
]


def _run(env_extra):
    env = {k: v for k, v in os.environ.items() if not (k.startswith("HDM_") or k.startswith("HDSDP_MI355X_"))}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "switch_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (env_extra, (r.stdout + r.stderr)[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("SWITCH_WORKER_JSON ")]
    assert line, (r.stdout + r.stderr)[-2000:]
    return json.loads(line[-1][len("SWITCH_WORKER_JSON "):])


@pytest.fixture(scope="module")
def default_run():
    return _run({})


@pytest.mark.parametrize("setting", SETTINGS, ids=[",".join(f"{k}={v}" for k, v in s.items()) for s in SETTINGS])
def test_switch_in_its_other_position_gives_the_same_numbers(setting, default_run):
    got = _run(setting)
    for inst, ref in default_run.items():
        cur = got[inst]
        for key, rv in ref.items():
            if key == "path":
                if "HDSDP_MI355X_FORCE_GEMM" not in setting and "SWITCH_WORKER_SHARDS" not in setting:
                    assert cur[key] == rv, (inst, key)
                continue
            if key == "sparse":
                assert cur[key] == (rv and setting.get("HDSDP_MI355X_SPARSE_KKT") != "0"), (inst, setting)
                continue
            a, b = np.asarray(cur[key], dtype=np.float64), np.asarray(rv, dtype=np.float64)
            tol = 1e-8 if key.startswith("step") else 1e-10
            den = max(float(np.max(np.abs(b))), 1e-300)
            assert float(np.max(np.abs(a - b))) / den <= tol, (inst, key, setting, float(np.max(np.abs(a - b))) / den)
