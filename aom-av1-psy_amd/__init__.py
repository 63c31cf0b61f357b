"""aom-av1-psy_amd: MI355X (gfx950) back end for the aom_dsp encoder hot path.

The product is the C-ABI shared library `lib/libaomhip.so` (HIP kernels + host
runtime, sources in `csrc/`, interface in `include/aomhip.h`).  This Python package
is only the thin ctypes binding used by tests and bench.py, plus the synthetic
frame / work-list generators of SURVEY.md section 8(d).  It has no CPU fallback:
importing `capi` without the built library raises.
"""
from . import capi, partition, synth  # noqa: F401

__all__ = ["capi", "partition", "synth"]
