"""ptamd: Python-side plumbing for the MI355X ray-queue path tracer.

  layout  -- numpy mirrors of the C-ABI structs (include/ptamd.h)
  host    -- ctypes binding of the C++ host scene library (BVH builders, scene graph, camera)
  device  -- ctypes binding of the HIP C-ABI (no CPU fallback)
  scenes  -- procedural scenes for the BASELINE configurations
  build   -- in-tree builds of the two shared libraries
"""
from . import layout  # noqa: F401
