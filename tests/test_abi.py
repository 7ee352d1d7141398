"""CPU: the C-ABI library loads and exports every symbol include/ams.h declares (no compute calls)."""
import ctypes
import os
import pytest

from ams_hip import _lib


def test_header_parses():
    protos = _lib.parse_header()
    assert 'ams_gemm_f32' in protos and 'ams_blstm_recurrent_fwd' in protos and len(protos) >= 20
    ret, args = protos['ams_front_conv_fwd']
    assert ret is ctypes.c_int32 and len(args) == 11


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(lib, name), name
    assert _lib.load().ams_abi_version() == 1


def test_ops_refuse_cpu_tensors():
    torch = pytest.importorskip('torch')
    from ams_hip import ops, AmsError
    with pytest.raises(AmsError):
        ops.front_filter(torch.zeros(4), torch.zeros(4, 2))
