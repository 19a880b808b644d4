"""CPU-side: libherald_ps.so loads and exports every libps name include/herald_ps.h declares."""
import ctypes
import os

from herald_amd import _lib


def test_libps_names_are_exported(lib):
    assert os.path.exists(_lib.PS_LIB_PATH)
    P = ctypes.CDLL(_lib.PS_LIB_PATH)
    names = _lib.declared_symbols(_lib.PS_HEADER_PATH)
    for must in ("InitTensor", "SparsePull", "SparsePush", "SSPushPull", "Wait", "SaveParam", "LoadParam", "rank", "nrank"):
        assert must in names
    missing = [n for n in names if not hasattr(P, n)]
    assert not missing, missing
    assert P.nrank() == 1 and P.rank() == 0          # no device access
