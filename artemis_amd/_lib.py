"""ctypes binding of libartemis_synth.so (the C ABI declared in include/artemis_synth.h).

The library is the product: if it is missing or does not load, every op raises -- there is no
CPU or eager-PyTorch fallback (the reference has none either: cupy_module/synth.py:310-311).
"""
from __future__ import annotations

import ctypes
import os

from . import build as _build

_lib = None

ABI_VERSION = 1
RADIUS_DEFAULT = -1
RADIUS_NONE = 0
RADIUS_AUTO = -2
RADIUS_SMALL = -3
GIN_ATOMIC = 0
GIN_DETERMINISTIC = 1
FRAME_F32 = 0
FRAME_BF16 = 1

_vp = ctypes.c_void_p
_i = ctypes.c_int

# name -> (restype, argtypes); one entry per declaration in include/artemis_synth.h
SIGNATURES = {
    "artemis_synth_forward": (_i, [_vp] * 5 + [_i] * 7 + [_vp]),
    "artemis_synth_backward": (_i, [_vp] * 9 + [_i] * 8 + [_vp, ctypes.c_size_t, _vp]),
    "artemis_synth_backward_workspace_bytes": (ctypes.c_size_t, [_i] * 7),
    "artemis_synth_forward_ex": (_i, [_vp] * 5 + [_i] * 12 + [_vp]),
    "artemis_synth_backward_ex": (_i, [_vp] * 9 + [_i] * 13 + [_vp, ctypes.c_size_t, _vp]),
    "artemis_synth_backward_workspace_bytes_ex": (ctypes.c_size_t, [_i] * 9),
    "artemis_synth_backward_noalloc": (_i, [_vp] * 9 + [_i] * 13 + [_vp]),
    "artemis_synth_head_forward": (_i, [_vp, _i] + [_vp] * 5 + [_i] * 9 + [_vp]),
    "artemis_synth_head_backward": (_i, [_vp, _i] + [_vp] * 9 + [_i] * 9 + [_vp]),
    "artemis_synth_offset_radius": (_i, [_vp, _vp, ctypes.c_size_t, _vp, _vp]),
    "artemis_synth_set_small_grid_tiles": (ctypes.c_longlong, [ctypes.c_longlong]),
    "artemis_synth_auto_query": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "artemis_synth_auto_reset": (None, []),
    "artemis_synth_set_gin_gather": (_i, [_i]),
    "artemis_synth_version": (ctypes.c_char_p, []),
    "artemis_synth_last_error": (ctypes.c_char_p, []),
}


class SynthLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("ARTEMIS_SYNTH_LIB", _build.LIB_PATH)


def load():
    """Load the shared library once per process; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise SynthLibraryError(
                f"{path} not found: build it with `python -m artemis_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:  # e.g. libamdhip64 missing
            raise SynthLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().artemis_synth_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (hipError {code}): {msg}")


def version() -> str:
    return load().artemis_synth_version().decode()
