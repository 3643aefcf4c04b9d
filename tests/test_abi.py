"""C-ABI checks that need no GPU: the library loads, exports every symbol that
include/artemis_synth.h declares, and its host-side validation behaves (no kernel is launched
by any call here: each returns before the first HIP call)."""
import ctypes
import os
import re

import pytest

from artemis_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "artemis_synth.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(artemis_synth_\w+)\s*\(", text)))


def test_library_is_built_in_tree():
    assert os.path.exists(build.LIB_PATH), "run python -m artemis_amd.build"
    assert os.path.dirname(build.LIB_PATH).startswith(ROOT)


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for want in ("artemis_synth_forward", "artemis_synth_backward",
                 "artemis_synth_backward_workspace_bytes", "artemis_synth_offset_radius",
                 "artemis_synth_version", "artemis_synth_last_error"):
        assert want in names


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.lib_path())
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in artemis_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_version_string():
    v = _lib.version()
    assert v.startswith("artemis_synth abi %d gfx950" % _lib.ABI_VERSION)
    abi = int(re.search(r"#define ARTEMIS_SYNTH_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert abi == _lib.ABI_VERSION


def test_header_constants_match_binding():
    text = open(HEADER).read()
    get = lambda n: int(re.search(r"#define %s \(?(-?\d+)\)?" % n, text).group(1))
    assert get("ARTEMIS_SYNTH_RADIUS_DEFAULT") == _lib.RADIUS_DEFAULT
    assert get("ARTEMIS_SYNTH_RADIUS_NONE") == _lib.RADIUS_NONE
    assert get("ARTEMIS_SYNTH_GIN_ATOMIC") == _lib.GIN_ATOMIC
    assert get("ARTEMIS_SYNTH_GIN_DETERMINISTIC") == _lib.GIN_DETERMINISTIC


def test_workspace_bytes():
    lib = _lib.load()
    f = lib.artemis_synth_backward_workspace_bytes
    # atomic mode: just the 256-byte header (fixed-point scale), and only for the LDS-staged shape
    assert f(2, 3, 6, 7, 5, 1, _lib.GIN_ATOMIC) == 256
    assert f(1, 2, 5, 9, 3, 2, _lib.GIN_ATOMIC) == 0
    # header (256 B) + one int64 per gradInput element
    assert f(2, 3, 6, 7, 5, 1, _lib.GIN_DETERMINISTIC) == 256 + 2 * 3 * 10 * 11 * 8
    assert f(1, 3, 1080, 1920, 5, 1, _lib.GIN_DETERMINISTIC) == 256 + 3 * 1084 * 1924 * 8
    assert f(1, 2, 5, 9, 3, 2, _lib.GIN_DETERMINISTIC) == 256 + 2 * 9 * 13 * 8
    assert f(0, 3, 6, 7, 5, 1, _lib.GIN_DETERMINISTIC) == 0


@pytest.mark.parametrize("dims,needle", [
    ((1, 3, 4, 4, 0, 1), "filter size"),
    ((1, 3, 4, 4, 5, 0), "dilation"),
    ((-1, 3, 4, 4, 5, 1), "negative"),
    ((70000, 3, 4, 4, 5, 1), "65535"),
    ((1, 3, 50000, 50000, 5, 1), "plane too large"),
])
def test_forward_rejects_bad_dimensions(dims, needle):
    lib = _lib.load()
    rc = lib.artemis_synth_forward(1, 1, 1, 1, 1, *dims, _lib.RADIUS_DEFAULT, None)
    assert rc == 1  # hipErrorInvalidValue
    assert needle in lib.artemis_synth_last_error().decode()
    with pytest.raises(RuntimeError, match=needle):
        _lib.check(rc, "artemis_synth_forward")


def test_null_pointers_are_rejected_before_any_launch():
    lib = _lib.load()
    assert lib.artemis_synth_forward(None, None, None, None, None, 1, 3, 4, 4, 5, 1, -1, None) == 1
    assert "NULL" in lib.artemis_synth_last_error().decode()
    assert lib.artemis_synth_backward(None, None, None, None, None, None, None, None, None,
                                      1, 3, 4, 4, 5, 1, 0, -1, None, 0, None) == 1
    assert lib.artemis_synth_backward(1, 1, 1, 1, 1, None, None, None, None,
                                      1, 3, 4, 4, 5, 1, 7, -1, None, 0, None) == 1
    assert "gin_mode" in lib.artemis_synth_last_error().decode()
    assert lib.artemis_synth_offset_radius(None, None, 10, None, None) == 1


def test_empty_problems_are_a_successful_noop():
    lib = _lib.load()
    for dims in ((0, 3, 4, 4, 5, 1), (1, 3, 0, 4, 5, 1), (1, 3, 4, 0, 5, 1), (1, 0, 4, 4, 5, 1)):
        assert lib.artemis_synth_forward(None, None, None, None, None, *dims, -1, None) == 0
    # backward with nothing requested and no pixels
    assert lib.artemis_synth_backward(None, None, None, None, None, None, None, None, None,
                                      0, 3, 4, 4, 5, 1, 0, -1, None, 0, None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("ARTEMIS_SYNTH_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.SynthLibraryError, match="no fallback"):
        _lib.load()


def test_ex_workspace_bytes_and_validation():
    lib = _lib.load()
    f = lib.artemis_synth_backward_workspace_bytes_ex
    # grad_input is (N, C, in_H, in_W): unpadded 6 x 7 frame
    assert f(2, 3, 6, 7, 5, 1, 6, 7, _lib.GIN_DETERMINISTIC) == 256 + 2 * 3 * 6 * 7 * 8
    assert f(2, 3, 6, 7, 5, 1, 10, 11, _lib.GIN_DETERMINISTIC) == \
        lib.artemis_synth_backward_workspace_bytes(2, 3, 6, 7, 5, 1, _lib.GIN_DETERMINISTIC)
    assert f(2, 3, 6, 7, 5, 1, 6, 7, _lib.GIN_ATOMIC) == 256
    # validation returns before any HIP call
    rc = lib.artemis_synth_forward_ex(None, None, None, None, None, 1, 3, 4, 4, 5, 1, 0, 4, 0, 0,
                                      0, -1, None)
    assert rc == 1 and "empty frame" in lib.artemis_synth_last_error().decode()
    rc = lib.artemis_synth_forward_ex(None, None, None, None, None, 1, 3, 4, 4, 5, 1, 4, 4,
                                      1 << 25, 0, 0, -1, None)
    assert rc == 1 and "origin" in lib.artemis_synth_last_error().decode()
    rc = lib.artemis_synth_forward_ex(None, None, None, None, None, 1, 3, 4, 4, 5, 1, -4, 4, 0, 0,
                                      0, -1, None)
    assert rc == 1
    # empty output is a successful no-op whatever the frame
    assert lib.artemis_synth_forward_ex(None, None, None, None, None, 0, 3, 4, 4, 5, 1, 4, 4, 0, 0,
                                        0, -1, None) == 0
    rc = lib.artemis_synth_backward_ex(None, None, None, None, None, None, None, None, None, 1, 3,
                                       4, 4, 5, 1, 4, 4, -2, -2, 0, 7, -1, None, 0, None)
    assert rc == 1 and "gin_mode" in lib.artemis_synth_last_error().decode()


def test_frame_dtype_validation():
    lib = _lib.load()
    # unknown dtype code
    rc = lib.artemis_synth_forward_ex(None, None, None, None, None, 1, 3, 4, 4, 5, 1, 4, 4, -2, -2,
                                      5, -1, None)
    assert rc == 1 and "frame_dtype" in lib.artemis_synth_last_error().decode()
    # bf16 frames only for the model's k = 5, d = 1, C = 3
    rc = lib.artemis_synth_forward_ex(None, None, None, None, None, 1, 2, 4, 4, 3, 1, 6, 6, 0, 0,
                                      _lib.FRAME_BF16, -1, None)
    assert rc == 1 and "bf16" in lib.artemis_synth_last_error().decode()
    rc = lib.artemis_synth_backward_ex(None, None, None, None, None, None, None, None, None, 1, 3,
                                       4, 4, 5, 2, 12, 12, 0, 0, _lib.FRAME_BF16, 0, -1, None, 0, None)
    assert rc == 1 and "bf16" in lib.artemis_synth_last_error().decode()
