# -*- coding: utf-8 -*-
"""BSS-eval SDR / SIR / SAR on MI355X -- host mirror of the reference's utils/bss_eval.py GPU entry point
(`bss_eval_sources_cupy`, utils/bss_eval.py:586-637, called by experiments/evaluation/eval.py:48-73).

The arithmetic runs in libams_bss.so (include/ams_bss.h: hipFFT + hipSOLVER + hand-written assembly / reduction kernels,
float64).  There is no CPU fallback: without the library or a GPU the functions raise.  Only the permutation choice over the
nsrc x nsrc criteria (a 2..6-element loop, :613-620) is host arithmetic, as in the reference.
"""
import ctypes
import itertools
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', 'ams_hip', 'libams_bss.so'))
FLEN = 512                      # utils/bss_eval.py:608

_lib = None
_ctx = {}                       # (nsrc, nsampl, flen, device) -> (ctx pointer, workspace tensor)


class BssError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BssError('libams_bss.so not found at %s -- run __graft_entry__.build() (no CPU fallback)' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.ams_bss_abi_version.restype = ctypes.c_int
        lib.ams_bss_create.restype = ctypes.c_int
        lib.ams_bss_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.ams_bss_destroy.restype = None
        lib.ams_bss_destroy.argtypes = [ctypes.c_void_p]
        lib.ams_bss_workspace_bytes.restype = ctypes.c_size_t
        lib.ams_bss_workspace_bytes.argtypes = [ctypes.c_void_p]
        lib.ams_bss_eval_pairs.restype = ctypes.c_int
        lib.ams_bss_eval_pairs.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        if lib.ams_bss_abi_version() != 1:
            raise BssError('libams_bss.so ABI version mismatch')
        _lib = lib
    return _lib


def _context(nsrc, nsampl, flen, device):
    key = (nsrc, nsampl, flen, str(device))
    if key not in _ctx:
        lib = _load()
        p = ctypes.c_void_p()
        st = lib.ams_bss_create(ctypes.byref(p), nsrc, nsampl, flen)
        if st != 0:
            raise BssError('ams_bss_create failed: %d' % st)
        nb = lib.ams_bss_workspace_bytes(p)
        ws = torch.empty((nb + 7) // 8, dtype=torch.float64, device=device)
        _ctx[key] = (p, ws, nb)
    return _ctx[key]


def bss_eval_pairs(reference_sources, estimated_sources, flen=FLEN):
    """[nsrc, nsampl] x2 (array-like or tensors) -> (sdr, sir, sar) numpy float64 [nsrc(jest), nsrc(jtrue)] pair matrices."""
    if not torch.cuda.is_available():
        raise BssError('bss_eval needs a GPU (there is no CPU fallback)')
    dev = reference_sources.device if torch.is_tensor(reference_sources) and reference_sources.is_cuda else torch.device('cuda')
    ref = torch.as_tensor(reference_sources).to(device=dev, dtype=torch.float64)
    est = torch.as_tensor(estimated_sources).to(device=dev, dtype=torch.float64)
    nsampl = est.shape[-1]
    ref = ref.reshape(-1, nsampl).contiguous()
    est = est.reshape(-1, nsampl).contiguous()
    nsrc = est.shape[0]
    if ref.shape != est.shape:
        raise BssError('reference and estimated sources must have the same shape, got %s and %s' % (tuple(ref.shape), tuple(est.shape)))
    p, ws, nb = _context(nsrc, nsampl, flen, dev)
    crit = torch.empty((3, nsrc, nsrc), dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    st = _load().ams_bss_eval_pairs(p, ref.data_ptr(), est.data_ptr(), crit.data_ptr(), info.data_ptr(), ws.data_ptr(), nb,
                                    torch.cuda.current_stream().cuda_stream)
    if st != 0:
        raise BssError('ams_bss_eval_pairs failed: %d' % st)
    c = crit.cpu().numpy()
    return c[0], c[1], c[2]


def bss_eval_sources_cupy(reference_sources, estimated_sources, compute_permutation=True, nsrc=2):
    """Same contract as the reference function of this name (utils/bss_eval.py:586-637): returns
    (sdr, sir, sar, perm) with estimated source perm[j] matched to true source j by best mean SIR."""
    sdr, sir, sar = bss_eval_pairs(np.asarray(reference_sources).reshape(nsrc, -1) if not torch.is_tensor(reference_sources)
                                   else reference_sources.reshape(nsrc, -1),
                                   np.asarray(estimated_sources).reshape(nsrc, -1) if not torch.is_tensor(estimated_sources)
                                   else estimated_sources.reshape(nsrc, -1))
    dum = np.arange(nsrc)
    if not compute_permutation:
        return sdr[dum, dum], sir[dum, dum], sar[dum, dum], dum
    perms = list(itertools.permutations(list(range(nsrc))))
    mean_sir = np.empty(len(perms))
    for i, perm in enumerate(perms):
        mean_sir[i] = np.mean(sir[list(perm), dum])
    popt = perms[int(np.argmax(mean_sir))]
    idx = (list(popt), dum)
    return sdr[idx], sir[idx], sar[idx], np.asarray(popt)


# the reference exposes the same metric under two names (numpy and cupy back ends); both map to the HIP path here
bss_eval_sources = bss_eval_sources_cupy
