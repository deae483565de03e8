"""Run-time switches of libivg from Python: the IVG_* environment variables listed in csrc/switches.h are read by the library when
it is loaded, at every engine construction and on ``ivg_reload_switches``.  ``set`` / ``override`` change them for engines that
already exist: they take effect at the next launch -- also under IVG_GRAPH=1, whose captured step graphs are keyed by the generation
of the switch table.  Process-global and not thread-safe: never call them while batches are in flight on other host threads."""
import contextlib
import os

from . import _lib

# Decode-GEMM footprint for SEVERAL batches in flight on one GPU (bench.py --lanes, INTEGRATION.md "streams"): with the whole LDS of
# a CU per workgroup (the default, fastest for one batch alone) the decode GEMMs of one batch lock the other batches' kernels out of
# the CU for their whole duration; at <= 52 KiB three fit, the q/k/v / gate-up / down GEMMs fall to the 4-wave second-generation
# kernel, and four batches in flight reach 5,680 instead of 5,350 predicted frames/s (profiles/r04_lanes.txt).
# This is the PROCESS-WIDE default; the per-engine form is ``LlamaForCausalLM(..., decode_lds_kb=40)`` / ``.set_decode_lds_kb(40)``
# (ivg_config.decode_lds_kb), which is what bench.py's lanes use: a latency engine and throughput engines can share a process.
# The budget is best effort (GEMMs whose smallest plan is larger keep it: lm_head) and it selects the kernel generation, so tokens
# produced under two budgets are each deterministic and batch-invariant but NOT bit-comparable with one another.
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
