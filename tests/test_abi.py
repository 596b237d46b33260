"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
public header declares (no compute calls -- there is no GPU here)."""
import os

import pytest

from daft_exprt import _hip as H


def test_library_exports_every_declared_symbol():
    if not os.path.exists(H.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = H.lib()
    protos = H.header_prototypes()
    assert len(protos) >= 4
    for name, _, _ in protos:
        assert hasattr(lib, name), f'{name} declared in include/daft_exprt_hip.h but not exported'
    assert lib.dx_abi_version() == 13
    assert lib.dx_last_error() is not None


def test_argument_errors_are_reported_not_ub():
    lib = H.lib()
    rc = lib.dx_conv1d(None, 0, 8, None, 1, None, None, 0, 8, None, 0, None, None, 1, 1, 8, 8, 3, 0, None)
    assert rc == -1 and b'null' in lib.dx_last_error()
    rc = lib.dx_conv1d(8, 0, 8, 8, 1, None, 8, 0, 8, None, 0, None, None, 1, 1, 12, 8, 3, 0, None)
    assert rc == -2 and b'multiples of 8' in lib.dx_last_error()
