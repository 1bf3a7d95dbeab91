"""quartetscores_amd -- MI355X-native quartet-support engine (hot path of lutteropp/QuartetScores).

The product path is hand-written HIP (quartetscores_amd/csrc) behind the C-ABI of
include/quartetscores_hip.h; this package is the thin Python host above it used by
tests/ and bench.py (torch supplies device memory, streams and torch.distributed).
There is no CPU fallback: importing the engine without the built library raises.
"""

__version__ = "0.1.0"
