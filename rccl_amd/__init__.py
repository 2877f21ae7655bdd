"""MI355X-native collective engine behind the RCCL/NCCL C API (all-reduce, reduce-scatter,
all-gather).  The compute lives in librccl_amd.so (hand-written HIP for gfx950); this package
is the thin host-side mirror of that C ABI used by the tests and the benchmark."""
from .nccl import *  # noqa: F401,F403
