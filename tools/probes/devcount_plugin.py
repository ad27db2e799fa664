"""pytest plugin (diagnostic): after every test, ask the HIP runtime for the device count and report the first test after
which it stops answering.  usage: python -m pytest -p tools.probes.devcount_plugin ..."""
import ctypes

_hip = None
_bad = False


def pytest_runtest_teardown(item, nextitem):
    global _hip, _bad
    if _bad:
        return
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(-1)
    rc = _hip.hipGetDeviceCount(ctypes.byref(n))
    if rc != 0 or n.value <= 0:
        _bad = True
        print("\n[devcount] after %s: hipGetDeviceCount rc=%d n=%d" % (item.nodeid, rc, n.value), flush=True)
