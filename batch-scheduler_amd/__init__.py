"""batch-scheduler_amd — MI355X-native gang-feasibility core for tenstack/batch-scheduler.

One hot path only: the PreFilter / Filter / Permit resource-fit arithmetic of the reference's
pkg/scheduler/core/core.go, as hand-written HIP for gfx950 behind the C ABI of include/bsched.h.
This package is the thin host side: SoA containers, the ctypes binding of libbsched.so, the
seeded synthetic snapshot generator and the build helper.  There is no CPU fallback: without the
HIP library and a GPU every compute entry point raises.
"""
from . import soa  # noqa: F401
from .soa import BatchOut, FitMasks, Groups, Nodes, Pods  # noqa: F401


def __getattr__(name):
    # lazy: keep `import batch-scheduler_amd.soa` usable without the shared library
    if name in ("Context", "BsError", "load_library", "LIB_PATH"):
        from . import capi
        return getattr(capi, name)
    if name in ("synth", "build", "capi", "plugin", "dist", "fitspec"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
