"""Run-time switches of libivg from Python: the IVG_* environment variables listed in csrc/switches.h are read by the library when
it is loaded, at every engine construction and on ``ivg_reload_switches``.  ``set`` / ``override`` change them for engines that
already exist (launch-policy switches such as the LDS budget of the decode GEMMs take effect at the next launch)."""
import contextlib
import os

from . import _lib

# Decode-GEMM footprint for SEVERAL batches in flight on one GPU (bench.py --lanes, INTEGRATION.md "streams"): with the whole LDS of
# a CU per workgroup (the default, fastest for one batch alone) the decode GEMMs of one batch lock the other batches' kernels out of
# the CU for their whole duration; at <= 52 KiB three fit, the q/k/v / gate-up / down GEMMs fall to the 4-wave second-generation
# kernel, and four batches in flight reach 5,680 instead of 5,350 predicted frames/s (profiles/r04_lanes.txt).
BATCHES_IN_FLIGHT = {"IVG_DECODE_LDS_KB": "40"}


def set(**kv):
    """``set(IVG_DECODE_LDS_KB=40, IVG_GRAPH=None)``: set / delete variables and publish them to the loaded library."""
    for k, v in kv.items():
        if not k.startswith("IVG_"):
            raise KeyError(k)
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    _lib.reload_switches()


@contextlib.contextmanager
def override(**kv):
    """Temporarily: ``with switches.override(**switches.BATCHES_IN_FLIGHT): ...``."""
    old = {k: os.environ.get(k) for k in kv}
    set(**kv)
    try:
        yield
    finally:
        set(**old)
