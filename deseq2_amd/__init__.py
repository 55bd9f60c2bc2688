"""deseq2_amd -- MI355X-native per-gene NB-GLM engine behind DESeq2's fitBeta / fitDisp /
fitDispGrid boundary.  The compute lives in libdeseq2_mi355x.so
(hand-written HIP for gfx950); this package is the host-side mirror of the reference's R
callers of that boundary."""
from . import _lib  # noqa: F401
from .native import fitBeta, fitDisp, fitDispGrid  # noqa: F401

__all__ = ["fitBeta", "fitDisp", "fitDispGrid"]
