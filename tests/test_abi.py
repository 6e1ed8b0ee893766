"""The C-ABI library loads and exports every symbol include/sisic.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sisic.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sisic_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from synt_isic_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_and_binding_agree(lib):
    from synt_isic_amd import _lib
    declared = _declared_symbols()
    assert declared, "no symbols parsed from include/sisic.h"
    assert sorted(_lib.SIGNATURES) == declared


def test_every_declared_symbol_is_exported(lib):
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/sisic.h but not exported"


def test_abi_version_and_error_channel(lib):
    assert lib.sisic_abi_version() == 3          # include/sisic.h: 3 since sisic_conv_args.fin_* / sisic_conv_finalizes (2: sisic_sample_frames, grown packed filters)
    # argument validation needs no GPU: NULL out-pointer is rejected with a message
    rc = lib.sisic_create(0, None)
    assert rc == -1
    assert b"sisic_create" in lib.sisic_last_error()
    assert lib.sisic_conv_packed_numel(64, 3, 3) == 16 * 9 * 64
    assert lib.sisic_conv_packed_numel(3, 64, 3) == 64 * 9 * 64
    assert lib.sisic_conv_packed_numel(768, 256, 1) == 256 * 768 * 7 // 2     # 1x1: the direct kernel's layout, the pointwise kernel's and the bf16x3 split behind it
    assert lib.sisic_conv_packed_numel(64, 64, 5) == -1


def test_struct_layout_matches_header():
    """ctypes mirrors of the two structs must have the C layout (sizes from the header's field list)."""
    from synt_isic_amd import _lib
    P, I, F = ctypes.sizeof(ctypes.c_void_p), ctypes.sizeof(ctypes.c_int), ctypes.sizeof(ctypes.c_float)
    assert P == 8 and I == 4
    assert ctypes.sizeof(_lib.ConvArgs) == 208          # 160 + the ABI-3 fin_* fields: 2 pointers, int, float, 3 pointers
    assert _lib.ConvArgs.fin_gamma.offset == 160 and _lib.ConvArgs.fin_groups.offset == 176 and _lib.ConvArgs.fin_eps.offset == 180 and _lib.ConvArgs.fin_scale.offset == 184 and _lib.ConvArgs.fin_mean_rstd.offset == 200
    assert _lib.ConvArgs.w_packed.offset == 48 and _lib.ConvArgs.out.offset == 128 and _lib.ConvArgs.w_winograd.offset == 144 and _lib.ConvArgs.stats_out.offset == 152
    assert ctypes.sizeof(_lib.UNetConfigC) == 4 * 4 + 3 * 32 + 4 * 4 + 8
