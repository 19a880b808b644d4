"""herald_amd -- MI355X-native embedding-access engine behind the Hetu/Herald operator API.

Only the hot path lives here: forward sparse gather, per-batch index plan (sorted-unique /
inverse / counts), backward dedup-reduce + fused sparse apply, the HET embedding cache, the laia
scheduler and the row-range sharded store.  The kernels are hand-written HIP for gfx950 in
csrc/, exported through the C-ABI of include/herald_amd.h; this package is the Python host side
that mirrors the reference's operator / plugin interfaces on top of it.
"""
from ._lib import HeraldAmdError, load  # noqa: F401

__version__ = "0.1.0"
