"""CPU: the C-ABI library loads and exports every symbol include/ams.h declares (no compute calls)."""
import ctypes
import os
import pytest

from ams_hip import _lib


def test_header_parses():
    protos = _lib.parse_header()
    assert 'ams_gemm_f32' in protos and 'ams_blstm_recurrent_fwd' in protos and len(protos) >= 20
    ret, args = protos['ams_front_conv_fwd']
    assert ret is ctypes.c_int32 and len(args) == 17


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(lib, name), name
    assert _lib.load().ams_abi_version() == _lib.ABI_VERSION


def test_library_exports_nothing_the_header_does_not_declare():
    """The product library carries no superseded experiment behind an undeclared symbol (round 3 shipped gemm_x3 / lstm_persist /
    ring_fwd_proj; they left the tree in round 4): every exported ams_* symbol is a prototype of include/ams.h."""
    import subprocess
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith('ams_'))
    declared = set(_lib.parse_header())
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_header_lists_every_environment_variable_the_library_reads():
    """State behind the ABI: the library's only process-wide inputs besides ams_gemm_set_arith are AMS_* variables read
    once; include/ams.h lists them all, and lists none the sources no longer read."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    for f in glob.glob(os.path.join(root, 'adaptive-multispeaker-separation_amd', 'csrc', '*.hip')) + \
            glob.glob(os.path.join(root, 'adaptive-multispeaker-separation_amd', 'csrc', '*.h')):
        read |= set(re.findall(r'getenv\("(AMS_[A-Z0-9_]+)"\)', open(f).read()))
    head = open(os.path.join(root, 'include', 'ams.h')).read()
    block = head[head.index('Environment read by the library'):head.index('#ifndef AMS_H')]
    listed = set(re.findall(r'\bAMS_[A-Z0-9_]+\b', block))
    assert read == listed, (sorted(read - listed), sorted(listed - read))


def test_ops_refuse_cpu_tensors():
    torch = pytest.importorskip('torch')
    from ams_hip import ops, AmsError
    with pytest.raises(AmsError):
        ops.front_filter(torch.zeros(4), torch.zeros(4, 2))


def test_ring_timeout_is_loud(monkeypatch):
    """A ring-recurrence launch that gave up a bounded wait sets the device's sticky error word (csrc/lstm_ring.hip,
    ops.ring_error_word).  bench.py and the tests call ops.raise_on_ring_errors() at their host sync, so such a step cannot be
    timed silently; the trainer repeats it on the per-step kernels instead (tests/test_gpu_ring_guard.py)."""
    torch = pytest.importorskip('torch')
    from ams_hip import ops, AmsError
    monkeypatch.setattr(ops, '_RING_ERR', {0: torch.zeros(1, dtype=torch.int32)})
    ops.raise_on_ring_errors()
    assert ops.persist_errors() == 0
    monkeypatch.setattr(ops, '_RING_ERR', {0: torch.ones(1, dtype=torch.int32)})
    assert ops.persist_errors() == 1
    with pytest.raises(AmsError):
        ops.raise_on_ring_errors()
    assert ops.persist_errors() == 0            # raising clears the word


def test_bss_library_exports_declared_symbols():
    """include/ams_bss.h: every declared entry point is exported (no compute call without a GPU)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'include', 'ams_bss.h')).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    names = set(re.findall(r'\b(ams_bss_\w+)\s*\(', src))
    assert names == {'ams_bss_abi_version', 'ams_bss_create', 'ams_bss_destroy', 'ams_bss_workspace_bytes', 'ams_bss_eval_pairs'}
    path = os.path.join(root, 'adaptive-multispeaker-separation_amd', 'ams_hip', 'libams_bss.so')
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.ams_bss_abi_version() == 1


def test_host_library_exports_declared_symbols():
    """include/ams_host.h: both helpers are exported and answer their known-answer checks (plain C: runs without a GPU)."""
    import re
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'include', 'ams_host.h')).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    names = set(re.findall(r'\b(ams_\w+)\s*\(', src))
    assert names == {'ams_crc32c', 'ams_mt_choice_rows'}
    lib = ctypes.CDLL(os.path.join(root, 'adaptive-multispeaker-separation_amd', 'ams_hip', 'libams_host.so'))
    for n in names:
        assert hasattr(lib, n), n
    lib.ams_crc32c.restype = ctypes.c_uint32
    lib.ams_crc32c.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    assert lib.ams_crc32c(b'123456789', 9) == 0xE3069283
    f = lib.ams_mt_choice_rows
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    key, pos, out = np.zeros(624, np.uint32), np.array([0], np.int32), np.zeros((2, 3), np.int32)
    assert f(key.ctypes.data, pos.ctypes.data, 2, 2, 3, out.ctypes.data) == -1            # C > l
    pos[0] = 625
    assert f(key.ctypes.data, pos.ctypes.data, 2, 5, 3, out.ctypes.data) == -1            # not a numpy stream position
