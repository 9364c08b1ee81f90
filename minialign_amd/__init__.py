"""minialign_amd -- MI355X-native seed-and-extend long-read alignment hot path (HIP kernels behind a C-ABI).

The compute lives in minialign_amd/libminialign_amd.so (built by __graft_entry__.build()); this package is the thin
Python mirror of the C-ABI declared in include/*.h.  There is no CPU fallback: loading fails loudly if the
library is missing, and the library itself refuses to initialise without a HIP device."""
import ctypes, os

_LIB = None

def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libminialign_amd.so')
        if not os.path.exists(p):
            raise RuntimeError('libminialign_amd.so not built; run `python __graft_entry__.py`')
        _LIB = ctypes.CDLL(p)
    return _LIB
