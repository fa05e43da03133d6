"""termdaw_amd -- MI355X-native offline render engine for termdaw audio graphs (hot path only).

`termdaw_amd.api` binds the C-ABI library (include/termdaw_amd.h); `termdaw_amd.workloads` holds the
synthetic BASELINE projects.  Nothing here falls back to a CPU implementation: if the HIP library
is missing or no GPU is present, rendering raises.
"""
__all__ = ["api", "workloads"]
