"""ctypes binding of libams_hip.so (the C ABI declared in include/ams.h).

There is NO CPU fallback: if the shared library is missing or a symbol is absent, importing the
compute ops raises.  Build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C adaptive-multispeaker-separation_amd/csrc``.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AMS_HIP_LIB') or os.path.join(_HERE, 'libams_hip.so')   # env: kernel-variant A/B runs
HEADER_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'include', 'ams.h'))

ABI_VERSION = 5            # include/ams.h: AMS_ABI_VERSION

_CT = {
    'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float, 'size_t': ctypes.c_size_t,
    'int32_t': ctypes.c_int32, 'ams_status': ctypes.c_int32, 'void': None,
}


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every prototype in include/ams.h."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(ams_status|size_t|int|long|void)\s+(ams_\w+)\s*\(([^)]*)\)\s*;', src):  # noqa: E501
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace('const ', '').split()[0]
                    argtypes.append(_CT[ty])
        protos[name] = (_CT[ret], argtypes)
    return protos


class AmsError(RuntimeError):
    pass


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # torch first: the ROCm wheel ships its own libamdhip64; loading ours afterwards makes the dynamic loader reuse that copy
    # (same SONAME).  The other order puts TWO HIP runtimes in the process and launches fail with hipErrorNoDevice.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AmsError('libams_hip.so not found at %s -- the HIP extension is required (no CPU fallback); '
                       'run __graft_entry__.build()' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, argtypes) in parse_header().items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise AmsError('libams_hip.so does not export %s (declared in include/ams.h)' % name)
        fn.restype = ret
        fn.argtypes = argtypes
    if lib.ams_abi_version() != ABI_VERSION:
        raise AmsError('libams_hip.so ABI version mismatch: the library is %d, this binding is %d -- rebuild (make -C csrc)'
                       % (lib.ams_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_STATUS = {-1: 'AMS_E_INVALID_ARG', -2: 'AMS_E_WORKSPACE_TOO_SMALL', -3: 'AMS_E_LAUNCH_FAILED'}


def check(status, what):
    if status != 0:
        extra = ''
        if status == -3:
            extra = ' (hipError %d)' % load().ams_last_error()
        raise AmsError('%s failed: %s%s' % (what, _STATUS.get(status, status), extra))
