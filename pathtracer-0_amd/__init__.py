"""pathtracer-0_amd — MI355X-native wavefront path tracer for the hot path of focksss/pathtracer-0.

Import name: `pathtracer_0_amd` (the directory name carries a hyphen; use `ptimport.load()` at
the repo root).  Contents: host-side scene producers (hostlib), synthetic workloads (scenes), the
HIP renderer behind the C ABI of include/pt_api.h (renderer), multi-GPU sharding helpers (shard).
"""
from . import hostlib, scenes  # noqa: F401
