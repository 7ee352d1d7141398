"""Tensor-level wrappers over the C ABI: torch provides device memory and the stream, nothing else.

Every function enqueues hand-written HIP kernels from libams_hip.so on torch's current stream.
All tensors must be contiguous fp32 CUDA(HIP) tensors unless stated.  There is no CPU path.
"""
import ctypes
import os as _os

import torch

from ._lib import load, check, AmsError

_vp = ctypes.c_void_p


def _p(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def _s():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise AmsError('ams_hip ops need device tensors (there is no CPU fallback)')
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise AmsError('ams_hip ops need contiguous fp32 tensors, got %s %s' % (t.dtype, tuple(t.stride())))


def _chk_rows(*ts):
    """Like _chk but rows may be strided (twin-interleaved BLSTM kernels): only the last dim must be dense."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise AmsError('ams_hip ops need device tensors (there is no CPU fallback)')
        if t.dtype != torch.float32 or t.stride(-1) != 1:
            raise AmsError('ams_hip ops need fp32 tensors with dense rows, got %s %s' % (t.dtype, tuple(t.stride())))


def _twin(a, b):
    """True when a and b are the two halves of one row-interleaved block [R, 2, C] (optim.FlatOptimizer twin layout)."""
    if a is None or b is None or a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
        return False
    C = a.shape[-1]
    if b.data_ptr() != a.data_ptr() + 4 * C or a.stride(-1) != 1 or b.stride(-1) != 1:
        return False
    return a.dim() == 1 or (a.stride(0) == 2 * C and b.stride(0) == 2 * C)


class _Profile(object):
    """Timing of individual launches on the launching stream (bench.py `roofline`).  Two modes: HIP events around a launch (eager
    launches), or device-clock STAMPS (ams_stamp: one-thread kernels writing the constant-rate wall clock) -- the only form that
    survives hipGraph capture: stamps recorded while a graph is being captured are replayed with it, so the durations read back
    are those of the launches INSIDE the replayed graph (each includes its two ~1.5 us kernel boundaries to the stamp kernels)."""

    def __init__(self):
        self.reset(False)

    def reset(self, enabled=False, stamps=False):
        self.enabled = enabled
        self.stamps = bool(stamps)
        self.records = []            # (start, stop, flops, bytes, tag, label): events, or stamp slots
        self.buf = torch.zeros(8192, dtype=torch.int64, device='cuda') if (enabled and stamps) else None
        self.nslots = 0

    def begin(self):
        if self.stamps:
            slot = self.nslots
            self.nslots += 2
            if self.nslots > self.buf.numel():
                raise AmsError('profile: more than %d stamped launches' % (self.buf.numel() // 2))
            check(load().ams_stamp(_p(self.buf), slot, _s()), 'ams_stamp')
            return slot
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, start, flops=0.0, nbytes=0.0, tag='', label=''):
        if self.stamps:
            check(load().ams_stamp(_p(self.buf), start + 1, _s()), 'ams_stamp')
            self.records.append((start, start + 1, flops, nbytes, tag, label))
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.records.append((start, e, flops, nbytes, tag, label))

    def tags(self):
        return sorted(set(r[4] for r in self.records))

    def summary(self, tag=None, label=None, prefix=None):
        torch.cuda.synchronize()
        ms = fl = by = 0.0
        n = 0
        ticks = self.buf.cpu().numpy() if self.stamps and self.buf is not None else None
        rate = float(load().ams_stamp_rate()) if self.stamps else 1.0
        for s, e, f, b, t, lab in self.records:
            if tag is not None and t != tag:
                continue
            if label is not None and lab != label:
                continue
            if prefix is not None and not t.startswith(prefix):
                continue
            ms += (float(ticks[e] - ticks[s]) / rate * 1e3) if self.stamps else s.elapsed_time(e)
            fl += f
            by += b
            n += 1
        return {'launches': n, 'ms': ms, 'flops': fl, 'bytes': by}


PROFILE = _Profile()


def _ws(nbytes, like):
    n = max(int(nbytes), 16)
    return torch.empty((n + 3) // 4, dtype=torch.float32, device=like.device)


_SK = {}
SK = _os.environ.get('AMS_GEMM_SK', '1') != '0'


def _sk(like):
    """(pointer, bytes) of the stream-K scratch of the current stream (include/ams.h: sk_scratch): one persistent buffer per device and
    stream -- launches that share one must be stream-ordered --, flags zeroed once (every launch leaves them zero)."""
    if not SK or not like.is_cuda:
        return _vp(0), 0
    cur = torch.cuda.current_stream()
    k = (like.device.index, cur.cuda_stream)
    t = _SK.get(k)
    if t is None:
        n = int(load().ams_gemm_sk_scratch_bytes())
        t = torch.empty((n + 3) // 4, dtype=torch.float32, device=like.device)
        t[:2048].zero_()
        _SK[k] = t
    return _vp(t.data_ptr()), t.numel() * 4


# ------------------------------------------------------------------ front
def front_filter(w, bases):
    _chk(w, bases)
    W, N = bases.shape
    f = torch.empty_like(bases)
    check(load().ams_front_filter_fwd(_p(w), _p(bases), _p(f), W, N, _s()), 'ams_front_filter_fwd')
    return f


def front_filter_bwd(w, bases, df):
    _chk(w, bases, df)
    W, N = bases.shape
    dw, db = torch.empty_like(w), torch.empty_like(bases)
    check(load().ams_front_filter_bwd(_p(w), _p(bases), _p(df), _p(dw), _p(db), W, N, _s()), 'ams_front_filter_bwd')
    return dw, db


def front_conv(x, f, hop, amax=None, measure=False):
    """amax = (bound of x, bound of f): fp16x3.  measure: the launch also leaves max |y| (tagged on y for the next product)."""
    _chk(x, f)
    Bt, L = x.shape
    W, N = f.shape
    T = -(-L // hop)
    y = torch.empty((Bt, T, N), dtype=torch.float32, device=x.device)
    lib = load()
    nb = lib.ams_front_conv_fwd_workspace_bytes(Bt, L, W, N, hop)
    ws = _ws(nb, x) if nb else None
    ay = None
    if measure and F16X3 and lib.ams_front_conv_fwd_measures_output(_p(x), _p(f), L, W, N, hop):
        ay = torch.empty(1, dtype=torch.float32, device=x.device)
    ev = PROFILE.begin() if PROFILE.enabled else None
    pa, pb, gt = _bounds(amax, ('front_conv', Bt, L, W, N, hop), ((x, Bt, L, L, 0), (f, W, N, N, 1)))
    skp, skn = _sk(x)
    check(lib.ams_front_conv_fwd(_p(x), _p(f), _p(y), Bt, L, W, N, hop, pa, pb, _p(ay), 0, _p(ws), nb, skp, skn, _s()),
          'ams_front_conv_fwd')
    if ay is not None:
        tag_amax(y, ay)
    if ev is not None:      # algorithmic bytes: waveform in, frames out, filter once
        PROFILE.end(ev, 2.0 * Bt * T * N * W, 4.0 * (Bt * L + Bt * T * N + W * N), gt + '<2,0>', 'front_conv')
    return y


def front_conv_bwd_filter(x, dy, W, hop):
    _chk(x, dy)
    Bt, L = x.shape
    N = dy.shape[2]
    lib = load()
    nb = lib.ams_front_conv_bwd_filter_workspace_bytes(Bt, L, W, N, hop)
    ws = _ws(nb, x)
    df = torch.empty((W, N), dtype=torch.float32, device=x.device)
    check(lib.ams_front_conv_bwd_filter(_p(x), _p(dy), _p(df), Bt, L, W, N, hop, _p(ws), nb, _s()), 'ams_front_conv_bwd_filter')
    return df


# ------------------------------------------------------------------ operand bounds of the fp16x3 products (include/ams.h: amax_a / amax_b)
F16X3 = _os.environ.get('AMS_GEMM_F16X3', '1') != '0'           # 0: nobody computes or passes bounds; every product stays bf16x6
_ONE = {}


def absmax(t, out=None):
    """1-element tensor holding max |t| (NaN if t has one): two small launches.  `out`: refresh an existing bound in place (so that
    captured graphs, which hold its address, see the new value)."""
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=t.device)
    _chk(t)
    check(load().ams_absmax_f32(_p(t), t.numel(), _p(out), _s()), 'ams_absmax_f32')
    return out


def amax_one(device):
    """The bound of everything a BLSTM layer emits: |tanh(c) * sigmoid(o)| < 1."""
    k = str(device)
    if k not in _ONE:
        _ONE[k] = torch.ones(1, dtype=torch.float32, device=device)
    return _ONE[k]


def tag_amax(t, a):
    """Producers that know a bound of their output attach it; consumers find it with amax_of()."""
    if F16X3 and a is not None:
        t._ams_amax = (a, t._version)                            # the tag dies with the first in-place write (autograd accumulation)
    return t


def amax_of(t):
    """The bound a producer attached to `t`, else max |t| computed now (None when fp16x3 is switched off)."""
    if not F16X3:
        return None
    b = t
    while b is not None:                                         # a view of a tagged tensor is bounded by the tag of its base
        a = getattr(b, '_ams_amax', None)
        if a is not None and a[1] == b._version:
            return a[0]
        b = b._base
    return absmax(t)


PASS = [0]                                                       # bumped by graph.Run: one evaluation pass = one measurement


def _frozen(*ts):
    """Every tensor carries the mark Network.freeze_weights() sets (inference recipes: nothing writes the variables between passes, and
    Network._weights_written() drops what was derived from them): bounds, gathered kernels and concatenated biases derived from such
    weights are kept ACROSS passes instead of being re-derived in every one."""
    return all(getattr(t, '_ams_frozen', False) for t in ts)


def drop_frozen_derivatives(t):
    for k in ('_ams_amax_cache', '_ams_wcat', '_ams_wcat_amax', '_ams_kbound', '_ams_bcat', '_ams_ps_w'):
        if hasattr(t, k):
            delattr(t, k)


class _ParamSource(object):
    """One optimizer's flat parameter buffer and the bound measured over it (shared by every variable the optimizer owns)."""
    __slots__ = ('flat', 'bound', 'seen', 'event', 'used', 'next', 'rolled', '__weakref__')

    def __init__(self, flat):
        self.flat = flat
        self.bound = torch.zeros(1, dtype=torch.float32, device=flat.device)
        self.next = torch.zeros(192, dtype=torch.int32, device=flat.device)    # include/ams.h: amax_slots of the optimizer kernels
        self.rolled = False             # an optimizer kernel keeps `bound` current (it leaves max |p| of what it writes there)
        self.seen = -1                  # the pass (PASS[0]) it was last measured in
        self.event = None               # measured on the side stream: recorded there, awaited by the first consumer
        self.used = -1                  # the last pass a product asked for this bound


_SOURCES = []                                                    # weak references: a source lives as long as its variables do


def param_amax(W):
    """Bound of a weight tensor, measured once in every evaluation pass (graph.Run) -- inside a captured step the measurement is
    part of the graph and is repeated by every replay, so no writer of weights (optimizer kernels, restore, `.data.copy_`) has to
    remember anything.  Variables owned by an optimizer share ONE bound over its whole flat buffer (a bound up to 2^10 above a
    tensor's own maximum costs that tensor no precision), measured by pass_begin() on the side stream while the front end runs;
    anything else is measured by itself at its first use."""
    if not F16X3:
        return None
    src = getattr(W, '_ams_amax_src', None)
    if src is not None:
        src.used = PASS[0]
        if src.seen != PASS[0]:                                  # not measured by pass_begin() (first use of this model): measure here
            absmax(src.flat, out=src.bound)
            src.seen, src.event = PASS[0], None
        elif src.event is not None:                              # measured on the side stream: every consuming stream waits once
            _await_pass_side()
        return src.bound
    c = getattr(W, '_ams_amax_cache', None)
    if c is None or (c[0] != PASS[0] and not _frozen(W)):
        W._ams_amax_cache = c = (PASS[0], absmax(W.detach(), out=c[1] if c is not None else None))
    return c[1]


class _PassSide(object):
    """What pass_begin() put on the side stream for the current pass, and which streams have already waited for it."""
    __slots__ = ('event', 'n', 'waited')

    def __init__(self):
        self.event, self.n, self.waited = None, -1, set()


_PASS_SIDE = _PassSide()


def _await_pass_side():
    """The current stream waits (once per pass and stream) for the side-stream work of pass_begin(): weight bounds, cleared ring
    sync buffers.  One cross-stream edge per pass, taken by whichever consumer comes first."""
    ps = _PASS_SIDE
    if ps.event is None or ps.n != PASS[0]:
        return
    cur = torch.cuda.current_stream()
    if cur.cuda_stream in ps.waited:
        return
    cur.wait_event(ps.event)
    ps.waited.add(cur.cuda_stream)


RING_ARENA = _os.environ.get('AMS_RING_ARENA', '1') != '0'


class _RingArena(object):
    """Sync buffers of the ring-recurrence launches of one pass, cleared TOGETHER: every ring launch needs its sync head zeroed, and a
    memset node in front of each of the six launches of a training step sat on the step's critical path (~5 us each).  Launch i of
    a pass takes slot i of one flat buffer that pass_begin() zeroes with ONE memset on the side stream, beside the input staging;
    the launch itself is told so (`safe` bit 2, include/ams.h) and is not preceded by anything.  A launch that finds no slot (first
    pass of a model, a larger shape, ops called outside a graph.Run) takes a private buffer and the per-launch memset as before, and
    leaves its size for the next pass.  A replaced flat buffer is kept alive: a captured hipGraph may hold its address."""

    def __init__(self):
        self.want = []              # bytes asked of slot i so far (maximum over passes)
        self.slots = []             # float32 views into self.flat
        self.flat = None
        self.retired = []
        self.next = 0               # slot the next ring launch of this pass takes
        self.cleared = -1           # the pass whose pass_begin() zeroed the slots
        self.used = -1              # the last pass a ring launch asked for a slot

    def begin(self, side_stream, device, zero=True):
        """Called by pass_begin() (inside `with torch.cuda.stream(side_stream)` when zero).  True when the slots are (being) cleared for this
        pass; zero=False: the caller clears self.flat itself."""
        self.next = 0
        if not self.want or self.used < PASS[0] - 2:
            return False
        have = [v.numel() * 4 for v in self.slots]
        if len(have) < len(self.want) or any(h < w for h, w in zip(have, self.want)):
            sizes = [(w + 255) // 256 * 256 for w in self.want]
            if self.flat is not None:
                self.retired.append(self.flat)
            self.flat = torch.empty(sum(sizes) // 4, dtype=torch.float32, device=device)
            self.slots, off = [], 0
            for n in sizes:
                self.slots.append(self.flat[off // 4:(off + n) // 4])
                off += n
        if zero:
            self.flat.zero_()
        self.cleared = PASS[0]
        return True

    def take(self, nbytes, like):
        """(sync buffer, `safe` bit) for the next ring launch."""
        i = self.next
        self.next += 1
        self.used = PASS[0]
        while len(self.want) <= i:
            self.want.append(0)
        self.want[i] = max(self.want[i], int(nbytes))
        if (self.cleared == PASS[0] and i < len(self.slots) and self.slots[i].numel() * 4 >= nbytes
                and self.slots[i].device == like.device):
            _await_pass_side()
            return self.slots[i], 4
        return _ws(nbytes, like), 0


_ARENAS = {}


def _ring_sync(nbytes, like):
    if not RING_ARENA or not like.is_cuda:
        return _ws(nbytes, like), 0
    k = like.device.index if like.device.index is not None else torch.cuda.current_device()
    ar = _ARENAS.get(k)
    if ar is None:
        ar = _ARENAS[k] = _RingArena()
    return ar.take(nbytes, like)


_DEFER_ZERO = []


def zero_deferred(t):
    """Zero `t` (an optimizer's flat gradient buffer) in the side-stream block of the NEXT pass_begin() instead of now: zeroing 47 MB
    in front of the forward pass cost the B = 64 step ~14 us of its critical path, and nothing writes a gradient before the backward
    pass.  The caller must call flush_deferred_zero() + await_pass_side() before its backward pass (models/network.py::_backward)."""
    if not t.is_cuda:
        t.zero_()
        return
    _DEFER_ZERO.append(t)


def flush_deferred_zero():
    """Zero, on the current stream, whatever no pass_begin() took since zero_deferred()."""
    while _DEFER_ZERO:
        _DEFER_ZERO.pop().zero_()


def await_pass_side():
    _await_pass_side()


ROLL_BOUNDS = _os.environ.get('AMS_ROLL_BOUNDS', '1') != '0'


def pass_begin(side_stream):
    """First node evaluation of a pass (graph.Node.value): work that depends on nothing the pass computes goes on the side stream,
    beside whatever the pass starts with (input staging, the front conv) -- the live optimizers' flat buffers are measured there
    instead of in front of the first product that needs the bound (47 MB: ~25 us), and the sync buffers of the pass's ring launches
    are zeroed there in one memset (_RingArena).  ONE event covers both; its first consumer waits for it (_await_pass_side).
    Inside a captured step this is part of the graph."""
    cur = torch.cuda.current_stream()
    dev = cur.device.index if cur.device.index is not None else torch.cuda.current_device()
    todo = []
    live = []
    for ref in _SOURCES if F16X3 else []:
        src = ref()
        if src is None:
            continue
        live.append(ref)
        if not src.flat.is_cuda or src.seen == PASS[0] or src.flat.device != cur.device or src.used < PASS[0] - 2:
            continue                                            # (only models whose bound a recent pass asked for)
        todo.append(src)
    if F16X3:
        _SOURCES[:] = live
    ar = _ARENAS.get(dev) if RING_ARENA else None
    if ar is not None:
        ar.next = 0
    zeros = [t for t in _DEFER_ZERO if t.device == cur.device]
    # sources whose bound the optimizer kernel keeps current (include/ams.h: bound_out) need nothing inside a captured step, where only
    # that kernel writes the weights between replays
    measure = [src for src in todo if not (src.rolled and ROLL_BOUNDS and torch.cuda.is_current_stream_capturing())]
    for src in todo:
        if src not in measure:
            src.seen, src.event = PASS[0], None
    if not measure and not zeros and (ar is None or not ar.want or ar.used < PASS[0] - 2):
        return
    if not measure:
        # nothing but clearing: ONE launch on the pass's own stream.  As a second root branch of a captured step the two fills cost
        # the replay a join of ~12 us in front of the first projection (profiles/r05_*), more than the ~12 us the launch takes.
        _DEFER_ZERO[:] = [t for t in _DEFER_ZERO if t.device != cur.device]
        arena = None
        if ar is not None and ar.begin(cur, cur.device, zero=False):
            arena = ar.flat
        regs = [t for t in zeros] + ([arena] if arena is not None else [])
        while regs:
            a = regs.pop(0)
            b = regs.pop(0) if regs else None
            ok = all(r is None or (r.is_contiguous() and r.data_ptr() % 16 == 0 and (r.numel() * r.element_size()) % 16 == 0) for r in (a, b))
            if ok:
                check(load().ams_zero2(_p(a), a.numel() * a.element_size(), _p(b), (b.numel() * b.element_size()) if b is not None else 0, _s()),
                      'ams_zero2')
            else:
                a.zero_()
                if b is not None:
                    b.zero_()
        _PASS_SIDE.event, _PASS_SIDE.n, _PASS_SIDE.waited = None, PASS[0], set()
        return
    side_stream.wait_stream(cur)
    with torch.cuda.stream(side_stream):
        for src in measure:
            absmax(src.flat, out=src.bound)
        for t in zeros:
            t.zero_()
        _DEFER_ZERO[:] = [t for t in _DEFER_ZERO if t.device != cur.device]
        if ar is not None:
            ar.begin(side_stream, cur.device)
        ev = torch.cuda.Event()
        ev.record(side_stream)
    for src in measure:
        src.seen, src.event = PASS[0], ev
    _PASS_SIDE.event, _PASS_SIDE.n, _PASS_SIDE.waited = ev, PASS[0], set()


def param_bounds_dirty():
    """Somebody other than an optimizer kernel wrote weights (checkpoint restore): a captured step, which takes the bound the optimizer
    kernel keeps, must not see a stale one -- measure every live source now."""
    for ref in _SOURCES:
        src = ref()
        if src is not None and src.flat.is_cuda:
            absmax(src.flat, out=src.bound)


def register_param_source(variables, flat):
    """FlatOptimizer: every variable it owns takes its bound from one measurement of the flat buffer."""
    import weakref
    src = _ParamSource(flat)
    flat._ams_src = src
    for v in variables:
        v._ams_amax_src = src
    _SOURCES.append(weakref.ref(src))
    return src


class _F16Audit(object):
    """Run-time guard of the fp16x3 products (include/ams.h: ams_range_share).  fp16x3 scales an operand by ONE power of two taken from
    its bound; entries below bound * 2^-17 keep fewer than 22 bits (absolute error bound * 2^-39 each).  While `active`, every product
    launched with bounds also measures, for both operands, how many non-zero entries lie down there and their energy; finish() turns that
    into the estimated relative error the lost bits add to the operand, r = sqrt(n_small) * bound * 2^-39 / |operand|_F; it also counts
    the operand's non-zero OUTER slices (rows of op(A), columns of op(B): what an output row / column is computed from) that lie
    entirely below the threshold -- an output row of such a slice is only accurate to ~2^-15 of its own scale (include/ams.h), however
    little energy it carries.  A product CLASS (entry point, shape, layouts) whose r exceeds `limit` (default 2^-22: the level of the f32
    arithmetic itself) or of whose outer slices more than `row_limit` (default 1 %) are out of range is DENIED: from then on _bounds()
    hands it NULL bounds, i.e. bf16x6, which needs no range.  Trainer.train audits one eager step every
    --f16_audit_every steps (models/network.py::train_audited) and re-captures its step graph when a class was denied."""

    def __init__(self):
        self.active = False
        self.records = []
        self.denied = set()
        self.limit = float(_os.environ.get('AMS_F16_AUDIT_LIMIT', 2.0 ** -22))
        self.row_limit = float(_os.environ.get('AMS_F16_AUDIT_ROWS', 0.01))
        self.report = []                # (key, operand 'A'/'B', n_small, energy share, r, share of outer slices out of range) of the last audit

    def begin(self):
        self.active, self.records = True, []

    def measure(self, key, role, t, rows, cols, ld, bound, outer=0):
        out = torch.zeros(3, dtype=torch.float32, device=t.device)
        check(load().ams_range_share(_p(t), rows, cols, ld, _p(bound), _p(out), _s()), 'ams_range_share')
        # outer slices entirely below the threshold (index glue of an audit that runs once in ~1000 steps, not a product path)
        v = torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset()).abs().amax(dim=1 - outer)
        slices = torch.stack([((v > 0) & (v < bound * 2.0 ** -17)).sum(), (v > 0).sum()]).to(torch.float32)
        self.records.append((key, role, out, bound, slices))

    def finish(self):
        """Host sync.  Returns the product classes newly denied."""
        self.active = False
        new, self.report = [], []
        for key, role, out, bound, slices in self.records:
            n_small, e_small, e_tot = [float(v) for v in out.cpu()]
            b = float(bound.cpu())
            r = (n_small ** 0.5) * b * 2.0 ** -39 / (e_tot ** 0.5) if e_tot > 0 else 0.0
            s_out, s_live = [float(v) for v in slices.cpu()]
            frac = s_out / s_live if s_live > 0 else 0.0
            self.report.append((key, role, n_small, e_small / e_tot if e_tot > 0 else 0.0, r, frac))
            if (r > self.limit or frac > self.row_limit) and key not in self.denied:
                self.denied.add(key)
                new.append(key)
        self.records = []
        return new


F16_AUDIT = _F16Audit()


def _bounds(amax, key=None, operands=None):
    """(pointer of A's bound, pointer of B's bound, profile tag prefix) for the product entry points: fp16x3 ('gemm16')
    when both bounds are there, else NULLs = bf16x6 / native f32 ('gemm').  key: the product class (F16_AUDIT may have denied it);
    operands: ((tensor, rows, cols, ld), (...)) of A and B as stored, measured while an audit is active."""
    if key is not None and key in F16_AUDIT.denied:
        return _vp(0), _vp(0), 'gemm'
    if F16X3 and amax is not None and amax[0] is not None and amax[1] is not None:
        if F16_AUDIT.active and key is not None and operands is not None:
            for role, (t, rows, cols, ld, outer), bnd in (('A', operands[0], amax[0]), ('B', operands[1], amax[1])):
                F16_AUDIT.measure(key, role, t, rows, cols, ld, bnd, outer)
        cur = torch.cuda.current_stream()
        amax[0].record_stream(cur)      # the launch may be on the side stream (functional.OVERLAP): the caching allocator must not
        amax[1].record_stream(cur)      # hand a bound's 4 bytes to a main-stream tensor while that product has yet to read them
        return _p(amax[0]), _p(amax[1]), 'gemm16'
    return _vp(0), _vp(0), 'gemm'


# Residency cap of the products launched next (functional.OVERLAP.cap): a launch ATTRIBUTE handed to every product entry point as
# its `lds_pad` argument (include/ams.h) -- Python-side state of the overlap scheduler, nothing behind the C ABI.
LDS_PAD = [0]


class lds_pad(object):
    """with ops.lds_pad(50000): ...  -- the products inside are launched residency-capped (beside a recurrence ring)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.old, LDS_PAD[0] = LDS_PAD[0], self.n

    def __exit__(self, *exc):
        LDS_PAD[0] = self.old



# ------------------------------------------------------------------ GEMM
def gemm(A, B, transA=False, transB=False, bias=None, out=None, accumulate=False, M=None, N=None, K=None,
         lda=None, ldb=None, ldc=None, mask=(0, 0), label='', amax=None):
    """out[M,N] (+)= op(A) op(B) (+ bias).  A/B may be 2-D tensors (dims inferred) or raw views with explicit
    M,N,K and leading dimensions (for column slices of wider buffers).  amax = (bound of A, bound of B): fp16x3 arithmetic."""
    lib = load()
    if M is None:
        _chk(A, B, bias)
        if transA:
            K_, M = A.shape
        else:
            M, K_ = A.shape
        if transB:
            N, K2 = B.shape
        else:
            K2, N = B.shape
        if K_ != K2:
            raise AmsError('gemm: inner dimensions differ (%d vs %d)' % (K_, K2))
        K = K_
        lda, ldb = A.stride(0), B.stride(0)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        ldc = N
    elif ldc is None:
        ldc = out.stride(0) if out.dim() == 2 else N
    for t_ in (A, B, out, bias):
        if t_ is not None and (t_.dtype != torch.float32 or not t_.is_cuda):
            raise AmsError('gemm: operands must be fp32 device tensors')
    pad = LDS_PAD[0]
    nb = lib.ams_gemm_workspace_bytes(M, N, K, 1, pad)
    ws = _ws(nb, A) if nb else None
    ev = PROFILE.begin() if PROFILE.enabled else None
    pa, pb, gt = _bounds(amax, ('gemm', label, M, N, K, bool(transA), bool(transB)),
                         ((A, K if transA else M, M if transA else K, lda, int(bool(transA))), (B, N if transB else K, K if transB else N, ldb, int(not transB))))
    skp, skn = _sk(A)
    check(lib.ams_gemm_f32(int(transA), int(transB), M, N, K, _p(A), lda, _p(B), ldb, _p(out), ldc, _p(bias), int(accumulate),
                           mask[0], mask[1], pa, pb, pad, _p(ws), nb, skp, skn, _s()), 'ams_gemm_f32')
    if ev is not None:
        PROFILE.end(ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N), gt + '<%d,%d>' % (int(bool(transA)), int(bool(transB))), label)
    return out


# ------------------------------------------------------------------ products from pre-split fp16 operand images (csrc/gemm_ps.hip)
PRESPLIT = _os.environ.get('AMS_GEMM_PRESPLIT', '1') != '0' and F16X3
# A captured training step is replayed both ways for a few steps and keeps the faster form (models/network.py::_train_graphed): the
# pre-split products run the matrix pipe at twice the duty of the in-product form, and on some boards that makes the clock governor
# settle ~8 % lower for the WHOLE step (2160 instead of 2350 MHz, tools/probes/step_power_probe.sh) -- more than the products save.
PS_AUTOTUNE = _os.environ.get('AMS_PS_AUTOTUNE', '1') != '0'
PS_LAUNCHES = [0]                                                # ams_gemm_ps launches issued (or captured) so far
PS_TUNED = {}                                                    # the last decision: {'presplit': bool, 'ms_presplit': .., 'ms_in_product': ..}


class presplit(object):
    """with ops.presplit(False): ...  -- the forward products inside take (or avoid) the pre-split form."""

    def __init__(self, on):
        self.on = bool(on) and F16X3

    def __enter__(self):
        global PRESPLIT
        self.old, PRESPLIT = PRESPLIT, self.on

    def __exit__(self, *exc):
        global PRESPLIT
        PRESPLIT = self.old


def ps_pack_rows(x2, amax, img=None):
    """PS32 image (include/ams.h) of x2 [R, K] (dense rows, any row pitch): uint8 [R, pitch]."""
    if x2.dtype != torch.float32 or not x2.is_cuda or x2.dim() != 2 or x2.stride(1) != 1:
        raise AmsError('ps_pack_rows: fp32 device matrix with dense rows expected')
    lib = load()
    R, K = x2.shape
    pitch = lib.ams_ps_image_pitch(K)
    if img is None:
        img = torch.empty((R, pitch), dtype=torch.uint8, device=x2.device)
    check(lib.ams_ps_pack_rows(_p(x2), x2.stride(0), _p(img), R, K, _p(amax), _s()), 'ams_ps_pack_rows')
    return img


def ps_pack_cols(w2, amax, img=None):
    """PS32 image of w2^T for w2 [K, N] (dense rows, any row pitch): uint8 [N, pitch] -- the B operand of x . w2."""
    if w2.dtype != torch.float32 or not w2.is_cuda or w2.dim() != 2 or w2.stride(1) != 1:
        raise AmsError('ps_pack_cols: fp32 device matrix with dense rows expected')
    lib = load()
    K, N = w2.shape
    pitch = lib.ams_ps_image_pitch(K)
    if img is None:
        img = torch.empty((N, pitch), dtype=torch.uint8, device=w2.device)
    check(lib.ams_ps_pack_cols(_p(w2), w2.stride(0), _p(img), K, N, _p(amax), _s()), 'ams_ps_pack_cols')
    return img


def gemm_ps(a_img, b_img, K, amax, bias=None, out=None, ldc=None, label=''):
    """out[M, N] = A . B^T (+ bias) from the two PS32 images (a_img [M, pitch], b_img [N, pitch]); amax = the bounds they were cut with."""
    lib = load()
    M, N = a_img.shape[0], b_img.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a_img.device)
        ldc = N
    elif ldc is None:
        ldc = out.stride(0) if out.dim() == 2 else N
    ev = PROFILE.begin() if PROFILE.enabled else None
    cur = torch.cuda.current_stream()
    amax[0].record_stream(cur)
    amax[1].record_stream(cur)
    check(lib.ams_gemm_ps(M, N, K, _p(a_img), _p(b_img), _p(out), ldc, _p(bias), _p(amax[0]), _p(amax[1]), _s()), 'ams_gemm_ps')
    PS_LAUNCHES[0] += 1
    if ev is not None:
        PROFILE.end(ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N), 'gemm16ps<0,1>', label)
    return out


def _ps_weight_image(W2, amax_w, owner):
    """PS32 image of W2^T, cut once per evaluation pass (inside a captured step: by every replay) and kept across passes for frozen
    weights (Network.freeze_weights; dropped with the other derived tensors by drop_frozen_derivatives): cached on `owner`."""
    c = getattr(owner, '_ams_ps_w', None)
    if (c is not None and c[1] == W2.data_ptr() and c[2] == tuple(W2.shape) and c[3] is amax_w and (c[0] == PASS[0] or _frozen(owner))):
        return c[4]
    img = ps_pack_cols(W2, amax_w)
    owner._ams_ps_w = (PASS[0], W2.data_ptr(), tuple(W2.shape), amax_w, img)
    return img


def forward_product_presplit(M, N, K, out, bias, amax, label, ldc):
    """Does out[M, N] = x . W (+ bias) run from pre-split operand images?  Both operand bounds at hand, fp16x3 not denied for this
    product class, no audit in progress, no residency cap, 16-byte addressable output; K may be anything (the images pad it)."""
    key = ('gemm', label, M, N, K, False, False)
    return (PRESPLIT and amax is not None and amax[0] is not None and amax[1] is not None and key not in F16_AUDIT.denied
            and not F16_AUDIT.active and LDS_PAD[0] == 0 and N % 4 == 0 and ldc % 4 == 0 and out.data_ptr() % 16 == 0
            and (bias is None or bias.data_ptr() % 16 == 0) and max(M, N) * ((K + 31) // 32 * 128) < 2 ** 31
            and load().ams_gemm_get_arith() != 0)


def forward_product(x2, W2, bias, out, amax, label, owner, ldc=None):
    """out[M, N] = x2 [M, K] . W2 [K, N] + bias: the forward products of the path (BLSTM input projection, Conv1D).  From pre-split
    operand images (csrc/gemm_ps.hip) where forward_product_presplit() says so -- x2's image is cut here, W2's once per pass (kept across
    passes for frozen weights); otherwise ams_gemm_f32."""
    M, K = x2.shape
    N = W2.shape[1]
    ldc = (out.stride(0) if out.dim() == 2 else N) if ldc is None else ldc
    if not (forward_product_presplit(M, N, K, out, bias, amax, label, ldc) and x2.stride(1) == 1 and W2.stride(1) == 1):
        return gemm(x2, W2, bias=bias, out=out, M=M, N=N, K=K, lda=x2.stride(0), ldb=W2.stride(0), ldc=ldc, label=label, amax=amax)
    a_img = ps_pack_rows(x2, amax[0])
    b_img = _ps_weight_image(W2, amax[1], owner)
    return gemm_ps(a_img, b_img, K, amax, bias=bias, out=out, ldc=ldc, label=label)


def gemm_at_b_colsum(A, B, out, bsum, accumulate=True, amax=None, ldc=None):
    """out[M,N] (+)= A^T B and bsum[N] (+)= column sums of B in ONE pass over B (A [K,M], B [K,N] row-major).  Returns False when
    the shapes / alignments do not allow the fused form (the caller then uses gemm + colsum)."""
    K, M = A.shape
    N = B.shape[1]
    ldc = out.stride(0) if ldc is None else ldc
    ok = (M % 4 == 0 and N % 4 == 0 and A.stride(0) % 4 == 0 and B.stride(0) % 4 == 0 and A.stride(1) == 1 and B.stride(1) == 1
          and out.stride(-1) == 1 and bsum.stride(-1) == 1 and bsum.data_ptr() % 16 == 0
          and all(t.data_ptr() % 16 == 0 for t in (A, B)) and _os.environ.get('AMS_GEMM_NOVEC') is None)
    if not ok:
        return False
    lib = load()
    pad = LDS_PAD[0]
    nb = lib.ams_gemm_workspace_bytes(M, N, K, 1, pad)
    ws = _ws(nb, A) if nb else None
    bws = _ws(32 * N * 4, A)
    ev = PROFILE.begin() if PROFILE.enabled else None
    pa, pb, gt = _bounds(amax, ('at_b_colsum', M, N, K), ((A, K, M, A.stride(0), 1), (B, K, N, B.stride(0), 1)))
    skp, skn = _sk(A)
    check(lib.ams_gemm_f32_at_b_colsum(M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), _p(out), ldc, int(accumulate),
                                       _p(bsum), int(accumulate), _p(bws), pa, pb, pad, _p(ws), nb, skp, skn, _s()),
          'ams_gemm_f32_at_b_colsum')
    if ev is not None:
        PROFILE.end(ev, 2.0 * M * N * K, 4.0 * (M * K + K * N + M * N), gt + '<1,0>', '')
    return True


def gemm_batched2(A0, A1, B0, B1, C0, C1, transA, transB, M, N, K, lda, ldb, ldc, accumulate=False, mask=(0, 0), amax=None):
    """Two products of one shape in ONE launch (operand pairs given as tensors/views; offsets taken from their addresses)."""
    for t_ in (A0, A1, B0, B1, C0, C1):
        if t_.dtype != torch.float32 or not t_.is_cuda:
            raise AmsError('gemm_batched2: operands must be fp32 device tensors')
    da, db_, dc = (A1.data_ptr() - A0.data_ptr()), (B1.data_ptr() - B0.data_ptr()), (C1.data_ptr() - C0.data_ptr())
    if da % 4 or db_ % 4 or dc % 4:
        raise AmsError('gemm_batched2: operand offsets must be whole floats')
    gemm_batched(A0, B0, C0, 2, da // 4, db_ // 4, dc // 4, transA, transB, M, N, K, lda, ldb, ldc, accumulate, amax, mask)


def gemm_batched(A, B, C, nbatch, a_zs, b_zs, c_zs, transA, transB, M, N, K, lda, ldb, ldc, accumulate=False, amax=None, mask=(0, 0)):
    """nbatch products of one shape in ONE launch: batch z reads A + z*a_zs, B + z*b_zs and writes C + z*c_zs (element strides)."""
    lib = load()
    for t_ in (A, B, C):
        if t_.dtype != torch.float32 or not t_.is_cuda:
            raise AmsError('gemm_batched: operands must be fp32 device tensors')
    pad = LDS_PAD[0]
    nb = lib.ams_gemm_workspace_bytes(M, N, K, nbatch, pad)
    ws = _ws(nb, A) if nb else None
    ev = PROFILE.begin() if PROFILE.enabled else None
    pa, pb, gt = _bounds(amax, ('batched', M, N, K, nbatch, bool(transA), bool(transB)),
                         ((A, K if transA else M, M if transA else K, lda, int(bool(transA))), (B, N if transB else K, K if transB else N, ldb, int(not transB))))
    skp, skn = _sk(A)
    check(lib.ams_gemm_f32_batched(int(transA), int(transB), M, N, K, _p(A), lda, a_zs, _p(B), ldb, b_zs, _p(C), ldc, c_zs, nbatch,
                                   int(accumulate), mask[0], mask[1], pa, pb, pad, _p(ws), nb, skp, skn, _s()),
          'ams_gemm_f32_batched')
    if ev is not None:
        PROFILE.end(ev, nbatch * 2.0 * M * N * K, nbatch * 4.0 * (M * K + K * N + M * N), gt + '<%d,%d>' % (int(bool(transA)), int(bool(transB))))


# ------------------------------------------------------------------ masks
def make_masks(rep_non_mix, B, S, a, b, take_abs, want_argmax=False, dpcl_E=None):
    """rep_non_mix: [B*S, ...] rows (b,s) row-major.  Returns Y [B, TF, S] (and int32 argmax [B, TF]).
    dpcl_E: the labels feed the fused deep-clustering loss on E-dimensional embeddings in this pass -- count them as they are written
    (include/ams.h: ams_dpcl_u_make_masks); dpcl_loss_fwd_u on the same Y then skips its own count launch."""
    _chk(rep_non_mix)
    TF = rep_non_mix.numel() // (B * S)
    Y = torch.empty((B, TF, S), dtype=torch.float32, device=rep_non_mix.device)
    if dpcl_E is not None and not want_argmax and S <= 8 and dpcl_E + S <= 64:
        lib = load()
        nb = lib.ams_dpcl_u_workspace_bytes(B, TF, dpcl_E, S)
        ws = _ws(nb, Y)
        check(lib.ams_dpcl_u_make_masks(_p(rep_non_mix), _p(Y), B, S, TF, dpcl_E, float(a), float(b), int(take_abs), _p(ws), nb, _s()),
              'ams_dpcl_u_make_masks')
        _DPCL_AHEAD[0] = (Y.data_ptr(), (B, TF, S, dpcl_E), PASS[0], ws)
        return Y
    am = torch.empty((B, TF), dtype=torch.int32, device=rep_non_mix.device) if want_argmax else None
    check(load().ams_make_masks(_p(rep_non_mix), _p(Y), _p(am), B, S, TF, float(a), float(b), int(take_abs), _s()), 'ams_make_masks')
    return (Y, am) if want_argmax else Y


# ------------------------------------------------------------------ BLSTM
import os as _os
# chain-per-XCD ring recurrence (csrc/lstm_ring.hip), the default: 1 = on, 0 = per-step kernels, 'safe' = on with the
# placement-independent (write-through) hand-off forced
LSTM_RING = _os.environ.get('AMS_LSTM_RING', '1')
_RING_ERR = {}                                                      # device index -> int32[1]: the sticky error word of every ring launch


def ring_error_word(device=None):
    """The sticky error word handed to every ring-recurrence launch on this device (include/ams.h: sticky_err): the kernels set it
    when a bounded in-launch wait gives up, nothing but ring_errors_clear() zeroes it.  One persistent buffer per device, so a
    replayed hipGraph keeps writing to the word the host reads; the fused optimizers take it as their skip guard."""
    dev = torch.cuda.current_device() if device is None else (device.index if hasattr(device, 'index') and device.index is not None
                                                               else torch.cuda.current_device())
    t = _RING_ERR.get(dev)
    if t is None:
        t = _RING_ERR[dev] = torch.zeros(1, dtype=torch.int32, device='cuda:%d' % dev)
    return t


_ERR_EXTRA = []                                                     # weak refs: words that carry OTHER ranks' ring errors (optim.FlatOptimizer)


def register_error_word(t):
    """A 1-element tensor that is non-zero when a PEER rank's ring launch gave up (the slot the data-parallel optimizer sends through
    its gradient all-reduce): ring_error_pending() reads it next to this device's own word, ring_errors_clear() zeroes it."""
    import weakref
    _ERR_EXTRA.append(weakref.ref(t))


def _err_words():
    live = [r for r in _ERR_EXTRA if r() is not None]
    _ERR_EXTRA[:] = live
    return list(_RING_ERR.values()) + [r() for r in live]


def ring_error_pending():
    """True when a ring launch since the last ring_errors_clear() abandoned a bounded wait -- on this rank, or (data parallel) on any
    rank as of the last gradient exchange (one host sync per word)."""
    return any(bool(t.item() != 0) for t in _err_words())


def ring_errors_clear():
    for t in _err_words():
        t.zero_()


def persist_errors():
    """Number of devices whose sticky ring error word is raised (host sync)."""
    return int(ring_error_pending())


def raise_on_ring_errors():
    """Fail loudly when a ring-recurrence launch gave up a bounded in-launch wait (csrc/lstm_ring.hip: its workgroups must all be
    resident at once; a co-running kernel that fills every CU's registers can keep some of them out past the wait limit).  The
    step that launch belonged to is invalid.  One host sync.  The trainer does not call this: it REPEATS such a step on
    the per-step kernels (utils/trainer.py::Trainer.train); benches and tests do, where a repeat would hide what they measure."""
    if ring_error_pending():
        ring_errors_clear()
        raise AmsError('a BLSTM ring recurrence launch abandoned a bounded wait: its workgroups were not all resident in time '
                       '(another kernel filled the CUs); the results of that step are invalid.  AMS_LSTM_RING=0 selects the '
                       'per-step recurrence kernels, which need no co-residency.')


def _padded_rows(x2, Dp):
    """[M, D] -> [M, Dp] with zero columns appended: an input width that is not a multiple of 4 (the STFT's F = 257 bins, 2F = 514 in
    the enhance stack) leaves its rows not 16-byte addressable, and every product reading them falls back to the scalar-load native-f32
    kernel (csrc/gemm.hip) -- 2-3x slower than the same product on the 16-bit pipe.  The copy is M * Dp * 4 bytes (5 MB at F = 257)."""
    xp = torch.zeros((x2.shape[0], Dp), dtype=x2.dtype, device=x2.device)
    xp[:, :x2.shape[1]].copy_(x2)
    return xp


def blstm_fwd(x, Kf, bf, Kb, bb, amax=None):
    """One BLSTM layer (utils/ops.py:358-383).  x [B,T,D]; K* [D+H,4H]; b* [4H].
    Returns out [B,T,2H] and the tensors the backward needs (G = activated gates, cst = cell states).
    amax: (bound of x, bound of the kernels) -> the input projection and the ring's recurrent product run as fp16x3."""
    _chk(x, bf, bb)
    _chk_rows(Kf, Kb)
    lib = load()
    B, T, D = x.shape
    H = Kf.shape[1] // 4
    ldu = Kf.stride(0)
    if Kb.stride(0) != ldu:
        raise AmsError('blstm: the two direction kernels must share one row stride')
    x2 = x.view(B * T, D)
    frozen = _frozen(Kf, Kb, bf, bb)
    if frozen:                                                   # inference recipes: gathered once, not in every pass
        c = getattr(Kf, '_ams_wcat', None)
        if c is None or c[0] is not Kb or c[1] != D:
            Kf._ams_wcat = c = (Kb, D, blstm_wcat(Kf, Kb, D))
        Wcat = c[2]
    else:
        Wcat = blstm_wcat(Kf, Kb, D)
    u_amax = amax[1] if amax is not None else None               # the caller's weight bound covers the recurrent kernels too
    if amax is None and F16X3 and x.is_cuda:
        if frozen:
            # ONE bound over both whole kernels, measured once: the projection AND the ring's recurrent product run as fp16x3
            c = getattr(Kf, '_ams_kbound', None)
            if c is None or c[0] is not Kb:
                Kf._ams_kbound = c = (Kb, torch.maximum(absmax(Kf), absmax(Kb)))
            amax, u_amax = (amax_of(x), c[1]), c[1]
        else:
            # no optimizer, hence no common bound of the two kernels: the gathered [D, 8H] matrix is measured once per pass
            c = getattr(Kf, '_ams_wcat_amax', None)
            if c is None or c[0] != PASS[0]:
                Kf._ams_wcat_amax = c = (PASS[0], absmax(Wcat, out=c[1] if c is not None else None))
            amax = (amax_of(x), c[1])
    if _twin(bf, bb):
        bias = torch.as_strided(bf, (8 * H,), (1,))
    elif frozen:
        c = getattr(bf, '_ams_bcat', None)
        if c is None or c[0] is not bb:
            bf._ams_bcat = c = (bb, torch.cat([bf, bb]))
        bias = c[1]
    else:
        bias = torch.cat([bf, bb])
    G = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=x.device)
    # hoisted input projection of BOTH directions as ONE MFMA GEMM [B*T, D] x [D, 8H]: the two [D,4H] halves of the TF
    # kernels are gathered side by side (a 2 x D x 4H copy) so N = 8H gives 19 x 40 = 760 tiles = 2.97 per CU instead of
    # two launches of 400 (1.56 per CU, i.e. 22 % of the CU-time idle).
    nring = lib.ams_blstm_ring_sync_bytes(B, H, 0) if LSTM_RING != '0' else 0
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=x.device)
    # ring recurrence: plane 0 = c_t (what every backward reads as `cst`), plane 1 = tanh(c_t) for the backward ring
    cst = torch.empty(((2, B, T, 2, H) if nring else (B, T, 2, H)), dtype=torch.float32, device=x.device)
    Dp = (D + 3) // 4 * 4
    if Dp != D and x.is_cuda and not forward_product_presplit(B * T, 8 * H, D, G, bias, amax, 'blstm_input_gemm', 8 * H):
        # rows of D floats are not 16-byte addressable: project zero-padded copies (x: +3 zero columns, kernels: +3 zero rows)
        Wp = torch.zeros((Dp, 8 * H), dtype=Wcat.dtype, device=Wcat.device)
        Wp[:D].copy_(Wcat)
        gemm(_padded_rows(x2, Dp), Wp, bias=bias, out=G, M=B * T, N=8 * H, K=Dp, lda=Dp, ldb=8 * H, ldc=8 * H, label='blstm_input_gemm',
             amax=amax)
    else:
        forward_product(x2, Wcat, bias, G.view(B * T, 8 * H), amax, 'blstm_input_gemm', Kf)
    if nring:
        sync, pre0 = _ring_sync(nring, x)
        check(lib.ams_blstm_ring_fwd(_p(G), _p(out), _p(cst[0]), _p(cst[1]), _p(Kf[D:]), _p(Kb[D:]), ldu,
                                     _p(u_amax if F16X3 else None), _p(sync), nring, _p(ring_error_word(x.device)), B, T, H,
                                     int(LSTM_RING == 'safe') | pre0, _s()), 'ams_blstm_ring_fwd')
        return out, G, cst
    pack = torch.empty(lib.ams_blstm_pack_floats(H, 0), dtype=torch.float32, device=x.device)
    check(lib.ams_blstm_recurrent_fwd(_p(G), _p(out), _p(cst), _p(Kf[D:]), _p(Kb[D:]), ldu, _p(pack), B, T, H, _s()),
          'ams_blstm_recurrent_fwd')
    return out, G, cst


def dense_fwd(x, W, b, amax=None):
    """u = x.W + b over the last axis (utils/ops.py:486-503)."""
    x2 = x.reshape(-1, x.shape[-1])
    out = torch.empty((x2.shape[0], W.shape[1]), dtype=torch.float32, device=x.device)
    return forward_product(x2, W, b, out, amax, '', W).view(x.shape[:-1] + (W.shape[1],))


def blstm_wcat(Kf, Kb, D):
    """[D, 8H] = input parts of the forward | backward kernels side by side: a zero-copy view when the two kernels are
    stored twin-interleaved (training through FlatOptimizer), else memcpy-class glue (2 x D x 4H floats)."""
    H4 = Kf.shape[1]
    if _twin(Kf, Kb):
        return torch.as_strided(Kf, (D, 2 * H4), (2 * H4, 1))
    Wcat = torch.empty((D, 2 * H4), dtype=torch.float32, device=Kf.device)
    Wcat[:, :H4].copy_(Kf[:D])
    Wcat[:, H4:].copy_(Kb[:D])
    return Wcat


def blstm_bwd_recurrent(x, Kf, Kb, G, cst, dout, amax_u=None):
    """BPTT recurrence of one BLSTM layer: overwrites G (activated gates) with d pre-activation.  Returns None, or (ring
    recurrence) dbpart [B, 2, 4H] = sum over t of d pre-activation, which blstm_bwd_weights turns into the bias gradients with a
    column sum over B rows instead of B*T.
    amax_u: bound of the recurrent kernels (what blstm_fwd took as amax[1]) -> the ring's recurrent product runs as fp16x3."""
    _chk(x, G, cst, dout)
    _chk_rows(Kf, Kb)
    lib = load()
    B, T, D = x.shape
    H = Kf.shape[1] // 4
    ldu = Kf.stride(0)
    # cst with a leading plane axis = written by the forward ring (plane 1 = tanh(c_t)); a 4-D cst came from the step kernels
    nring = lib.ams_blstm_ring_sync_bytes(B, H, 1) if (LSTM_RING != '0' and cst.dim() == 5) else 0
    if nring:
        sync, pre0 = _ring_sync(nring, x)
        dbpart = torch.empty((B, 2, 4 * H), dtype=torch.float32, device=x.device)
        check(lib.ams_blstm_ring_bwd(_p(G), _p(cst[0]), _p(cst[1]), _p(dout), _p(dbpart), _p(Kf[D:]), _p(Kb[D:]), ldu,
                                     _p(amax_u if F16X3 else None), _p(sync), nring, _p(ring_error_word(x.device)), B, T, H,
                                     int(LSTM_RING == 'safe') | pre0, _s()), 'ams_blstm_ring_bwd')
        tag_amax(G, sync.view(-1)[2:3])                          # max |dZ| came out of the same launch (float word 2 of the sync head)
        return dbpart
    pack = torch.empty(lib.ams_blstm_pack_floats(H, 1), dtype=torch.float32, device=x.device)
    dc = torch.empty((B, 2, H), dtype=torch.float32, device=x.device)
    check(lib.ams_blstm_recurrent_bwd(_p(G), _p(cst), _p(dout), _p(dc), _p(Kf[D:]), _p(Kb[D:]), ldu, _p(pack), B, T, H, _s()),
          'ams_blstm_recurrent_bwd')


# ---- DropoutWrapper(cell, keep, keep, keep) around each direction (utils/ops.py:363,373,379; --recurrent_dropout != 0, training)
def blstm_dropout_masks(B, T, D, H, keep, device, generator=None):
    """The eight keep-masks of one BLSTM layer, scaled by 1/keep (tf.nn.dropout): 'in' [2,B,T,D] (one per direction wrapper),
    'h', 'c' [B,T,2,H] on the two parts of the carried state (TF 1.4 maps the state dropout over the whole LSTMStateTuple),
    'out' [B,T,2H] on the cell outputs.  Drawn on the device: TensorFlow's stream is not reproducible, the distribution is."""
    def m(*shape):
        return (torch.rand(*shape, device=device, generator=generator) < keep).to(torch.float32) / keep
    # AMS_RDROPOUT_MASK_C=0: leave the cell state unmasked -- what DropoutWrapper does in the TensorFlow releases that have
    # `dropout_state_filter_visitor` (default: h only); 1.4.0, the release of the reference's containers, masks both parts
    mc = m(B, T, 2, H) if _os.environ.get('AMS_RDROPOUT_MASK_C', '1') != '0' else torch.ones(B, T, 2, H, device=device)
    return {'in': m(2, B, T, D), 'h': m(B, T, 2, H), 'c': mc, 'out': m(B, T, 2 * H)}


def blstm_fwd_dropout(x, Kf, bf, Kb, bb, masks):
    """One BLSTM layer under the reference's dropout wrappers.  Returns the layer output [B,T,2H] and what the backward needs."""
    _chk(x, bf, bb, masks['in'], masks['h'], masks['c'], masks['out'])
    _chk_rows(Kf, Kb)
    lib = load()
    B, T, D = x.shape
    H = Kf.shape[1] // 4
    ldu = Kf.stride(0)
    if Kb.stride(0) != ldu:
        raise AmsError('blstm: the two direction kernels must share one row stride')
    M = B * T
    xd = x.unsqueeze(0) * masks['in']                            # [2,B,T,D]: elementwise glue of a default-off flag
    G = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=x.device)
    Gv = G.view(-1)
    # the two wrappers see different inputs: one projection per direction, written into its half of the [B*T, 8H] rows
    gemm(xd[0].reshape(M, D), Kf, bias=bf, out=Gv, M=M, N=4 * H, K=D, lda=D, ldb=ldu, ldc=8 * H)
    gemm(xd[1].reshape(M, D), Kb, bias=bb, out=Gv[4 * H:], M=M, N=4 * H, K=D, lda=D, ldb=ldu, ldc=8 * H)
    out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=x.device)
    hs = torch.empty_like(out)
    cst = torch.empty((B, T, 2, H), dtype=torch.float32, device=x.device)
    cs = torch.empty_like(cst)
    pack = torch.empty(lib.ams_blstm_pack_floats(H, 0), dtype=torch.float32, device=x.device)
    check(lib.ams_blstm_recurrent_fwd_dropout(_p(G), _p(out), _p(cst), _p(hs), _p(cs), _p(masks['h']), _p(masks['c']), _p(Kf[D:]),
                                              _p(Kb[D:]), ldu, _p(pack), B, T, H, _s()), 'ams_blstm_recurrent_fwd_dropout')
    return out * masks['out'], (xd, G, cst, cs, hs)


def blstm_bwd_dropout(dy, x, Kf, Kb, saved, masks, need_dx=True):
    """BPTT of blstm_fwd_dropout.  DESTROYS G.  Returns dx, dKf, dbf, dKb, dbb."""
    xd, G, cst, cs, hs = saved
    _chk(dy, G, cst, cs, hs)
    lib = load()
    B, T, D = x.shape
    H = Kf.shape[1] // 4
    ldu = Kf.stride(0)
    M = B * T
    dout = (dy * masks['out']).contiguous()
    pack = torch.empty(lib.ams_blstm_pack_floats(H, 1), dtype=torch.float32, device=x.device)
    dc = torch.empty((B, 2, H), dtype=torch.float32, device=x.device)
    check(lib.ams_blstm_recurrent_bwd_dropout(_p(G), _p(cst), _p(cs), _p(dout), _p(dc), _p(masks['h']), _p(masks['c']), _p(Kf[D:]),
                                              _p(Kb[D:]), ldu, _p(pack), B, T, H, _s()), 'ams_blstm_recurrent_bwd_dropout')
    dKf = torch.empty((D + H, 4 * H), dtype=torch.float32, device=x.device)
    dKb = torch.empty((D + H, 4 * H), dtype=torch.float32, device=x.device)
    dbf = torch.empty(4 * H, dtype=torch.float32, device=x.device)
    dbb = torch.empty(4 * H, dtype=torch.float32, device=x.device)
    dZ = G.view(-1)
    gemm(xd[0].reshape(M, D), dZ, transA=True, out=dKf, M=D, N=4 * H, K=M, lda=D, ldb=8 * H, ldc=4 * H)
    gemm(xd[1].reshape(M, D), dZ[4 * H:], transA=True, out=dKb, M=D, N=4 * H, K=M, lda=D, ldb=8 * H, ldc=4 * H)
    # recurrent kernels: the MASKED states paired with dZ one step on; biases: column sums of dZ
    blstm_bwd_weights(x, hs, G, dKf, dbf, dKb, dbb, False, part='u')
    blstm_bwd_weights(x, hs, G, dKf, dbf, dKb, dbb, False, part='bias')
    dx = None
    if need_dx:
        dxf = gemm(dZ, Kf, transB=True, M=M, N=D, K=4 * H, lda=8 * H, ldb=ldu, ldc=D,
                   out=torch.empty((M, D), dtype=torch.float32, device=x.device))
        dxb = gemm(dZ[4 * H:], Kb, transB=True, M=M, N=D, K=4 * H, lda=8 * H, ldb=ldu, ldc=D,
                   out=torch.empty((M, D), dtype=torch.float32, device=x.device))
        dx = dxf.view(B, T, D) * masks['in'][0] + dxb.view(B, T, D) * masks['in'][1]
    return dx, dKf, dbf, dKb, dbb


def blstm_bwd_dx(G, Kf, Kb, B, T, D, amax=None):
    """dx = dZ . [Wx_f | Wx_b]^T  (the only hoisted product on the backward critical path), one GEMM with K = 8H."""
    H = Kf.shape[1] // 4
    M = B * T
    Wcat = blstm_wcat(Kf, Kb, D)
    dx = torch.empty((B, T, D), dtype=torch.float32, device=G.device)
    gemm(G.view(-1), Wcat, transB=True, out=dx, M=M, N=D, K=8 * H, lda=8 * H, ldb=8 * H, ldc=D, amax=amax)
    return dx


def blstm_bwd_weights(x, out, G, dKf, dbf, dKb, dbb, accumulate, part='all', dbpart=None, amax=None):
    """Weight gradients of one BLSTM layer from dZ (= G after blstm_bwd_recurrent), written (or accumulated) into the given
    buffers: dWx = x^T dZ, dU = h_prev^T dZ (time-shifted, masked at sequence boundaries), db = column sums.
    part: 'all' | 'wx' (input kernels + biases) | 'u' (recurrent kernels) -- the two halves are independent and may be
    issued on different streams.  amax: (bound of x, bound of out, bound of dZ) -> fp16x3 products."""
    lib = load()
    am_wx = (amax[0], amax[2]) if amax is not None else None
    am_u = (amax[1], amax[2]) if amax is not None else None
    B, T, D = x.shape
    H = dKf.shape[1] // 4
    M = B * T
    x2 = x.view(M, D)
    dZf = G.view(-1)                       # direction 0 columns start at 0, ld = 8H
    dZb = G.view(-1)[4 * H:]
    acc = bool(accumulate)
    _chk_rows(dKf, dKb)
    Dp = (D + 3) // 4 * 4
    if part in ('all', 'wx'):
        if Dp != D and x.is_cuda:
            # input width not a multiple of 4 (see _padded_rows): the product runs on a zero-padded copy of x into a [Dp, 8H] scratch,
            # whose first D rows are the gradient
            dWcat = gemm(_padded_rows(x2, Dp), dZf, transA=True, M=Dp, N=8 * H, K=M, lda=Dp, ldb=8 * H, ldc=8 * H,
                         out=torch.empty((Dp, 8 * H), dtype=torch.float32, device=x.device), amax=am_wx)
            if acc:
                dKf[:D].add_(dWcat[:D, :4 * H])
                dKb[:D].add_(dWcat[:D, 4 * H:])
            else:
                dKf[:D].copy_(dWcat[:D, :4 * H])
                dKb[:D].copy_(dWcat[:D, 4 * H:])
        elif _twin(dKf, dKb):
            # twin-interleaved gradient block: [dWx_f | dWx_b] IS a row-major [D, 8H] matrix -> written in place; when the two bias
            # gradients are adjacent too, db = column sums of dZ come out of the SAME pass over dZ (the product's tile_m == 0
            # workgroups add up the rows they stage anyway, finished inside the launch): no column-sum launches at all
            dWcat = torch.as_strided(dKf, (D, 8 * H), (8 * H, 1))
            fused = (_twin(dbf, dbb) and x2.is_cuda and
                     gemm_at_b_colsum(x2, G.view(M, 8 * H), dWcat, torch.as_strided(dbf, (8 * H,), (1,)), accumulate=acc, amax=am_wx, ldc=8 * H))
            if fused:
                part = {'all': 'u', 'wx': 'none'}[part]
            else:
                gemm(x2, dZf, transA=True, M=D, N=8 * H, K=M, lda=D, ldb=8 * H, ldc=8 * H, accumulate=acc, out=dWcat, amax=am_wx)
        else:
            dWcat = gemm(x2, dZf, transA=True, M=D, N=8 * H, K=M, lda=D, ldb=8 * H, ldc=8 * H,
                         out=torch.empty((D, 8 * H), dtype=torch.float32, device=x.device))
            if acc:
                dKf[:D].add_(dWcat[:, :4 * H])
                dKb[:D].add_(dWcat[:, 4 * H:])
            else:
                dKf[:D].copy_(dWcat[:, :4 * H])
                dKb[:D].copy_(dWcat[:, 4 * H:])
    if part in ('all', 'wx', 'bias'):
        if dbpart is not None and _twin(dbf, dbb):      # ring BPTT already summed dZ over time: column sum over B rows only
            nb = lib.ams_colsum_workspace_bytes(B, 8 * H)
            ws = _ws(nb, x)
            check(lib.ams_colsum(_p(dbpart), _p(dbf), B, 8 * H, 8 * H, int(acc), _p(ws), nb, _s()), 'ams_colsum')
        elif dbpart is not None:
            nb = lib.ams_colsum_workspace_bytes(B, 4 * H)
            ws = _ws(nb, x)
            check(lib.ams_colsum(_p(dbpart), _p(dbf), B, 4 * H, 8 * H, int(acc), _p(ws), nb, _s()), 'ams_colsum')
            ws2 = _ws(nb, x)
            check(lib.ams_colsum(_p(dbpart.view(-1)[4 * H:]), _p(dbb), B, 4 * H, 8 * H, int(acc), _p(ws2), nb, _s()), 'ams_colsum')
        elif _twin(dbf, dbb):                  # adjacent bias gradients: one column-sum over all 8H columns of dZ
            nb = lib.ams_colsum_workspace_bytes(M, 8 * H)
            ws = _ws(nb, x)
            check(lib.ams_colsum(_p(dZf), _p(dbf), M, 8 * H, 8 * H, int(acc), _p(ws), nb, _s()), 'ams_colsum')
        else:
            nb = lib.ams_colsum_workspace_bytes(M, 4 * H)
            ws = _ws(nb, x)
            check(lib.ams_colsum(_p(dZf), _p(dbf), M, 4 * H, 8 * H, int(acc), _p(ws), nb, _s()), 'ams_colsum')
            ws2 = _ws(nb, x)
            check(lib.ams_colsum(_p(dZb), _p(dbb), M, 4 * H, 8 * H, int(acc), _p(ws2), nb, _s()), 'ams_colsum')
    if part in ('all', 'u'):
        # forward dir pairs out[b,t-1] with dZ[b,t]; backward dir pairs out[b,t+1] with dZ[b,t]
        of = out.view(-1)
        if M - 1 <= 0:                         # a single (batch, time) row has no previous state: dU = 0
            if not acc:
                dKf[D:].zero_()
                dKb[D:].zero_()
        elif dKf.stride(0) == dKb.stride(0):
            # both directions in ONE launch: 2 x (3 x 10) tiles share the chip instead of queueing behind each other
            gemm_batched2(of, of[2 * H + H:], dZf[8 * H:], dZb, dKf[D:], dKb[D:], True, False, H, 4 * H, M - 1, 2 * H, 8 * H,
                          dKf.stride(0), accumulate=acc, mask=(T, T - 1), amax=am_u)
        else:
            gemm(of, dZf[8 * H:], transA=True, out=dKf[D:], accumulate=acc, M=H, N=4 * H, K=M - 1, lda=2 * H, ldb=8 * H,
                 ldc=dKf.stride(0), mask=(T, T - 1))
            gemm(of[2 * H + H:], dZb, transA=True, out=dKb[D:], accumulate=acc, M=H, N=4 * H, K=M - 1, lda=2 * H, ldb=8 * H,
                 ldc=dKb.stride(0), mask=(T, T - 1))


def blstm_bwd(x, Kf, Kb, out, G, cst, dout, need_dx=True, amax_u=None):
    """BPTT for one BLSTM layer.  DESTROYS G (it becomes d pre-activation).  Returns dx, dKf, dbf, dKb, dbb.
    amax_u: see blstm_bwd_recurrent."""
    _chk(x, out, G, cst, dout)
    B, T, D = x.shape
    H = Kf.shape[1] // 4
    dbpart = blstm_bwd_recurrent(x, Kf, Kb, G, cst, dout, amax_u=amax_u)
    dKf = torch.empty(Kf.shape, dtype=torch.float32, device=x.device)
    dKb = torch.empty(Kb.shape, dtype=torch.float32, device=x.device)
    dbf = torch.empty(4 * H, dtype=torch.float32, device=x.device)
    dbb = torch.empty(4 * H, dtype=torch.float32, device=x.device)
    blstm_bwd_weights(x, out, G, dKf, dbf, dKb, dbb, False, dbpart=dbpart)
    dx = blstm_bwd_dx(G, Kf, Kb, B, T, D) if need_dx else None
    return dx, dKf, dbf, dKb, dbb


# ------------------------------------------------------------------ l2norm / dense
def l2norm_fwd(u, E):
    _chk(u)
    rows = u.numel() // E
    v = torch.empty_like(u)
    inv = torch.empty(rows, dtype=torch.float32, device=u.device)
    check(load().ams_l2norm_fwd(_p(u), _p(v), _p(inv), rows, E, _s()), 'ams_l2norm_fwd')
    return v, inv


def l2norm2_fwd(u, E):
    """normalise(normalise(u)) over groups of E in one pass -> (xn, inv = 1/|u|, inv2 = 1/|normalise(u)|): the bits of two l2norm_fwd
    calls (which is what runs when the rows are not 16-byte addressable or E has no slab kernel)."""
    _chk(u)
    rows = u.numel() // E
    xn = torch.empty_like(u)
    inv = torch.empty(rows, dtype=torch.float32, device=u.device)
    inv2 = torch.empty(rows, dtype=torch.float32, device=u.device)
    st = load().ams_l2norm2_fwd(_p(u), _p(xn), _p(inv), _p(inv2), rows, E, _s())
    if st != 0:
        v, inv = l2norm_fwd(u, E)
        xn, inv2 = l2norm_fwd(v, E)
    return xn, inv, inv2


def l2norm_kmeans_normalize(u, E):
    """kmeans_normalize(l2norm_fwd(u)) in one pass, same bits (include/ams.h); the two calls where the fused kernel does not apply."""
    _chk(u)
    rows = u.numel() // E
    xn = torch.empty_like(u)
    st = load().ams_l2norm_kmeans_normalize(_p(u), _p(xn), rows, E, _s())
    if st != 0:
        v, _ = l2norm_fwd(u, E)
        xn = kmeans_normalize(v.view(-1, E)).view_as(u)
    return xn


def l2norm_bwd(v, inv, dv, E):
    _chk(v, inv, dv)
    du = torch.empty_like(v)
    check(load().ams_l2norm_bwd(_p(v), _p(inv), _p(dv), _p(du), inv.numel(), E, _s()), 'ams_l2norm_bwd')
    return du


def colsum_into(x2, out, accumulate):
    _chk(x2, out)
    lib = load()
    rows, cols = x2.shape
    nb = lib.ams_colsum_workspace_bytes(rows, cols)
    ws = _ws(nb, x2)
    check(lib.ams_colsum(_p(x2), _p(out), rows, cols, cols, int(bool(accumulate)), _p(ws), nb, _s()), 'ams_colsum')


def colsum(x2):
    _chk(x2)
    lib = load()
    rows, cols = x2.shape
    nb = lib.ams_colsum_workspace_bytes(rows, cols)
    ws = _ws(nb, x2)
    out = torch.empty(cols, dtype=torch.float32, device=x2.device)
    check(lib.ams_colsum(_p(x2), _p(out), rows, cols, cols, 0, _p(ws), nb, _s()), 'ams_colsum')
    return out


# ------------------------------------------------------------------ DPCL loss
def dpcl_loss_fwd(V, Y):
    """V [B,TF,E], Y [B,TF,S] -> out[4] = (cost, term1, term2, term3), workspace for the backward."""
    _chk(V, Y)
    lib = load()
    B, TF, E = V.shape
    S = Y.shape[2]
    nb = lib.ams_dpcl_workspace_bytes(B, TF, E, S)
    ws = _ws(nb, V)
    out = torch.empty(4, dtype=torch.float32, device=V.device)
    check(lib.ams_dpcl_loss_fwd(_p(V), _p(Y), _p(out), B, TF, E, S, _p(ws), nb, _s()), 'ams_dpcl_loss_fwd')
    return out, ws


_DPCL_AHEAD = [None]          # (Y.data_ptr(), (B, TF, S, E), PASS[0], workspace) of the latest dpcl_count_labels_ahead / make_masks(dpcl_E=)


def dpcl_count_labels_ahead(Y, E):
    """Label counts of the fused loss, issued before U exists (include/ams.h: ams_dpcl_u_count_labels); the dpcl_loss_fwd_u of the
    same pass on the same Y picks the workspace up and skips its own count launch."""
    _chk(Y)
    lib = load()
    B, TF, S = Y.shape
    nb = lib.ams_dpcl_u_workspace_bytes(B, TF, E, S)
    ws = _ws(nb, Y)
    check(lib.ams_dpcl_u_count_labels(_p(Y), B, TF, E, S, _p(ws), nb, _s()), 'ams_dpcl_u_count_labels')
    _DPCL_AHEAD[0] = (Y.data_ptr(), (B, TF, S, E), PASS[0], ws)
    return ws


def dpcl_loss_fwd_u(U, Y, want_V=False):
    """Fused l2-normalise + DPCL loss on the dense output U [B,TF,E] -> (out[4], inv [B,TF], V or None, ws)."""
    _chk(U, Y)
    lib = load()
    B, TF, E = U.shape
    S = Y.shape[2]
    nb = lib.ams_dpcl_u_workspace_bytes(B, TF, E, S)
    ahead, _DPCL_AHEAD[0] = _DPCL_AHEAD[0], None
    ready = ahead is not None and ahead[:3] == (Y.data_ptr(), (B, TF, S, E), PASS[0])
    ws = ahead[3] if ready else _ws(nb, U)
    out = torch.empty(4, dtype=torch.float32, device=U.device)
    inv = torch.empty(B, TF, dtype=torch.float32, device=U.device)
    V = torch.empty_like(U) if want_V else None
    ev = PROFILE.begin() if PROFILE.enabled else None
    check(lib.ams_dpcl_loss_fwd_u(_p(U), _p(Y), _p(inv), _p(V), _p(out), B, TF, E, S, int(ready), _p(ws), nb, _s()),
          'ams_dpcl_loss_fwd_u')
    if ev is not None:      # algorithmic bytes: read U and Y once, write 1/|u| (+ V when asked)   
        PROFILE.end(ev, 2.0 * B * TF * (E + S) * (E + S), 4.0 * B * TF * (E + S + 1 + (E if want_V else 0)), 'dpcl_gram_u', 'dpcl')
    return out, inv, V, ws


def dpcl_loss_bwd_u(U, Y, inv, ws, upstream=None):
    _chk(U, Y, inv, upstream)
    B, TF, E = U.shape
    S = Y.shape[2]
    d = torch.empty_like(U)
    ev = PROFILE.begin() if PROFILE.enabled else None
    check(load().ams_dpcl_loss_bwd_u(_p(U), _p(Y), _p(inv), _p(upstream), _p(d), B, TF, E, S, _p(ws), _s()),
          'ams_dpcl_loss_bwd_u')
    if ev is not None:      # read U, Y, 1/|u|; write dU
        PROFILE.end(ev, 2.0 * B * TF * (E + S) * E, 4.0 * B * TF * (2 * E + S + 1), 'dpcl_bwd_u', 'dpcl')
    if F16X3:
        # max |dU| came out of the same pass (include/ams.h: ams_dpcl_u_amax_offset): the bound of the two dense-layer products
        off = load().ams_dpcl_u_amax_offset(B, TF, E, S) // 4
        tag_amax(d, ws.view(-1)[off:off + 1])
    return d


def dpcl_loss_bwd(V, Y, ws, inv=None, upstream=None):
    _chk(V, Y, inv, upstream)
    B, TF, E = V.shape
    S = Y.shape[2]
    d = torch.empty_like(V)
    check(load().ams_dpcl_loss_bwd(_p(V), _p(Y), _p(inv), _p(upstream), _p(d), B, TF, E, S, _p(ws), _s()), 'ams_dpcl_loss_bwd')
    return d


# ------------------------------------------------------------------ optional input conditioning / weightings
_PRE = {None: 0, 'None': 0, 'none': 0, 'abs': 1, 'sqrt': 2, 'log': 3}
_NORM = {None: 0, 'None': 0, 'none': 0, '01': 1, 'meanstd': 2, 'silent': 3}
_WMODE = {None: 0, 'None': 0, 'none': 0, 'linear': 1, 'sqrt': 2, 'square': 3}


def row_transform(x, pre=None, norm=None, thr=0.0):
    """Per-utterance conditioning of X [B, ...] (network.py:409-454,504-521): pre in {abs, sqrt, log}, norm in {01, meanstd, silent}."""
    _chk(x)
    out = torch.empty_like(x)
    rows = x.shape[0]
    check(load().ams_row_transform(_p(x), _p(out), rows, x.numel() // rows, _PRE[pre], _NORM[norm], float(thr), _s()), 'ams_row_transform')
    return out


def weight_masks(X, y, mode=None, silence_thr=None):
    """y [B,TF,S] *= f(|X|/max|X|) and/or the silence-loss mask (network.py:381-396); returns a new tensor."""
    _chk(X, y)
    B, TF, S = y.shape
    out = y.clone()
    check(load().ams_weight_masks(_p(X), _p(out), B, TF, S, _WMODE[mode], int(silence_thr is not None),
                                  float(silence_thr if silence_thr is not None else 0.0), _s()), 'ams_weight_masks')
    return out


def silence_weights(lat, thr):
    """Kmeans_2.py:76-80: notsilent[b,l] = log10(max(lat[b]) / lat[b,l]) < thr."""
    _chk(lat)
    w = torch.empty_like(lat)
    rows = lat.shape[0]
    check(load().ams_silence_weights(_p(lat), _p(w), rows, lat.numel() // rows, float(thr), _s()), 'ams_silence_weights')
    return w


def pretrain_separator_fwd(y, B, S, separation):
    """adapt.py:173-196: y [B(1+S), T, N] -> [B*S, T, N]."""
    _chk(y)
    TN = y.numel() // y.shape[0]
    out = torch.empty((B * S,) + tuple(y.shape[1:]), dtype=torch.float32, device=y.device)
    check(load().ams_pretrain_separator_fwd(_p(y), _p(out), B, S, TN, 0 if separation == 'mask' else 1, _s()),
          'ams_pretrain_separator_fwd')
    return out


def pretrain_separator_bwd(dout, B, S, separation):
    _chk(dout)
    TN = dout.numel() // dout.shape[0]
    dy = torch.empty((B * (1 + S),) + tuple(dout.shape[1:]), dtype=torch.float32, device=dout.device)
    check(load().ams_pretrain_separator_bwd(_p(dout), _p(dy), B, S, TN, 0 if separation == 'mask' else 1, _s()),
          'ams_pretrain_separator_bwd')
    return dy


# ------------------------------------------------------------------ enhance output stage / L41 speaker vectors
_NONLIN = {None: 0, 'None': 0, 'none': 0, 'softmax': 1, 'tanh': 2}


def enhance_output_fwd(u, X, S, nonlinearity, want_separated=True):
    """u [B*S, T, F] (rows (b,s)), X [B, T, F] -> cost_in [B, TF, S], separated [B, S, TF] (network.py:640-660)."""
    _chk(u, X)
    B = X.shape[0]
    TF = X.numel() // B
    cost_in = torch.empty((B, TF, S), dtype=torch.float32, device=u.device)
    sep = torch.empty((B, S, TF), dtype=torch.float32, device=u.device) if want_separated else None
    check(load().ams_enhance_output_fwd(_p(u), _p(X), _p(cost_in), _p(sep), B, S, TF, _NONLIN[nonlinearity], _s()),
          'ams_enhance_output_fwd')
    return cost_in, sep


def enhance_output_bwd(u, X, S, nonlinearity, d_cost_in, d_sep):
    _chk(u, X, d_cost_in, d_sep)
    B = X.shape[0]
    TF = X.numel() // B
    du = torch.empty_like(u)
    check(load().ams_enhance_output_bwd(_p(u), _p(X), _p(d_cost_in), _p(d_sep), _p(du), B, S, TF, _NONLIN[nonlinearity], _s()),
          'ams_enhance_output_bwd')
    return du


def l41_speaker_fwd(table, I, normalize):
    """table [nspk,E], I int32 [B,S] -> vs [B,S,E] (L41.py:60-68)."""
    _chk(table)
    if I.dtype != torch.int32 or not I.is_cuda or not I.is_contiguous():
        raise AmsError('l41_speaker: I must be a contiguous int32 device tensor')
    nspk, E = table.shape
    R = I.numel()
    vs = torch.empty(tuple(I.shape) + (E,), dtype=torch.float32, device=table.device)
    check(load().ams_l41_speaker_fwd(_p(table), _p(I), _p(vs), R, E, nspk, int(bool(normalize)), _s()), 'ams_l41_speaker_fwd')
    return vs


def l41_speaker_bwd(table, I, d_vs, normalize):
    _chk(table, d_vs)
    nspk, E = table.shape
    d = torch.empty_like(table)
    check(load().ams_l41_speaker_bwd(_p(table), _p(I), _p(d_vs), _p(d), I.numel(), E, nspk, int(bool(normalize)), _s()),
          'ams_l41_speaker_bwd')
    return d


# ------------------------------------------------------------------ optimizers
def _guard(p, guard):
    """The optimizers' skip word: the device's sticky ring-error word unless the caller passes its own (or False for none)."""
    if guard is False:
        return None
    return ring_error_word(p.device) if guard is None else guard


def _slots_args(p):
    """(amax_slots, bound_out) of the optimizer entry points: the parameter source that owns flat buffer `p` gets its bound kept current."""
    src = getattr(p, '_ams_src', None)
    if src is None or not F16X3:
        return _vp(0), _vp(0)
    src.rolled = True
    return _p(src.next), _p(src.bound)


def opt_amsgrad(p, g, m, v, vhat, lr_t, beta1, beta2, eps, grad_scale=1.0, guard=None, scale_dev=None):
    _chk(p, g, m, v, vhat)
    check(load().ams_opt_amsgrad(_p(p), _p(g), _p(m), _p(v), _p(vhat), p.numel(), lr_t, beta1, beta2, eps, grad_scale,
                                 _p(_guard(p, guard)), *_slots_args(p), _p(scale_dev), _s()), 'ams_opt_amsgrad')


def opt_rmsprop(p, g, ms, lr, decay=0.9, eps=1e-10, grad_scale=1.0, guard=None, scale_dev=None):
    _chk(p, g, ms)
    check(load().ams_opt_rmsprop(_p(p), _p(g), _p(ms), p.numel(), lr, decay, eps, grad_scale, _p(_guard(p, guard)), *_slots_args(p), _p(scale_dev), _s()),
          'ams_opt_rmsprop')


def opt_momentum(p, g, acc, lr, momentum=0.9, grad_scale=1.0, guard=None, scale_dev=None):
    _chk(p, g, acc)
    check(load().ams_opt_momentum(_p(p), _p(g), _p(acc), p.numel(), lr, momentum, grad_scale, _p(_guard(p, guard)), *_slots_args(p), _p(scale_dev), _s()),
          'ams_opt_momentum')


_STAGE = {}


def stage_inputs(src, dst, src2=None, dst2=None, want_amax=True):
    """dst <- src (flat fp32, 16-byte aligned), dst2 <- src2 (any small contiguous pair), and a persistent 1-element tensor holding
    max |src| (None when not wanted / fp16x3 off): one launch (include/ams.h: ams_stage_inputs)."""
    lib = load()
    k = (dst.device.index, dst.data_ptr())
    st = _STAGE.get(k)
    if st is None:
        n = int(lib.ams_stage_inputs_scratch_bytes())
        st = _STAGE[k] = (torch.zeros(n // 4, dtype=torch.int32, device=dst.device), torch.zeros(1, dtype=torch.float32, device=dst.device))
    am = st[1] if (want_amax and F16X3) else None
    n2 = src2.numel() * src2.element_size() if src2 is not None else 0
    check(lib.ams_stage_inputs(_p(src), _p(dst), src.numel(), _p(src2), _p(dst2), n2, _p(am), _p(st[0]), _s()), 'ams_stage_inputs')
    return am


def clip_scale(ss, pre_scale, clip):
    """1-element device tensor clip / max(sqrt(ss) * pre_scale, clip) (tf.clip_by_global_norm, network.py:185-190): no host sync."""
    out = torch.empty(1, dtype=torch.float32, device=ss.device)
    check(load().ams_clip_scale(_p(ss), float(pre_scale), float(clip), _p(out), _s()), 'ams_clip_scale')
    return out


def sumsq(x):
    _chk(x)
    ws = _ws(4096, x)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(load().ams_sumsq(_p(x), _p(out), x.numel(), _p(ws), 4096, _s()), 'ams_sumsq')
    return out


# ------------------------------------------------------------------ pre-training regularisers (adapt.py:127-132, 310-316)
def abs_colsum(y2):
    """p_hat[M] = sum over rows of |y2[Bt, M]|."""
    _chk(y2)
    Bt, M = y2.shape
    lib = load()
    nb = lib.ams_abs_colsum_workspace_bytes(Bt, M)
    ws = _ws(nb, y2)
    out = torch.empty(M, dtype=torch.float32, device=y2.device)
    check(lib.ams_abs_colsum_fwd(_p(y2), _p(out), Bt, M, _p(ws), nb, _s()), 'ams_abs_colsum_fwd')
    return out


def kl_sparsity_fwd(p_hat, p):
    _chk(p_hat)
    out = torch.empty(1, dtype=torch.float32, device=p_hat.device)
    ws = _ws(4096, p_hat)
    check(load().ams_kl_sparsity_fwd(_p(p_hat), _p(out), p_hat.numel(), float(p), _p(ws), 4096, _s()), 'ams_kl_sparsity_fwd')
    return out


def kl_sparsity_bwd(y2, p_hat, upstream, p, gscale=1.0):
    """d (sum kl_div(p, p_hat)) / d y2 through p_hat = sum_b |y2|, times the device scalar `upstream` and the host scalar gscale."""
    _chk(y2, p_hat, upstream)
    Bt, M = y2.shape
    dy = torch.empty_like(y2)
    check(load().ams_kl_sparsity_bwd(_p(y2), _p(p_hat), _p(upstream), float(gscale), _p(dy), Bt, M, float(p), 0, _s()), 'ams_kl_sparsity_bwd')
    return dy


def negative_energy_fwd(y2):
    _chk(y2)
    Bt, M = y2.shape
    out = torch.empty(1, dtype=torch.float32, device=y2.device)
    ws = _ws(4096, y2)
    check(load().ams_negative_energy_fwd(_p(y2), _p(out), Bt, M, _p(ws), 4096, _s()), 'ams_negative_energy_fwd')
    return out


def sumsq_bwd(x, upstream, scale=1.0, mode=0):
    """upstream[0] * scale * 2x (mode 0) or * 2 min(x, 0) (mode 1)."""
    _chk(x, upstream)
    dx = torch.empty_like(x)
    check(load().ams_sumsq_bwd(_p(x), _p(upstream), float(scale), _p(dx), x.numel(), int(mode), 0, _s()), 'ams_sumsq_bwd')
    return dx


# ------------------------------------------------------------------ framed products / overlap-add (STFT, synthesis)
def frames_matmul(x, Bm, hop, T, pad_left):
    """out[(r,t), n] = sum_k xpad[r, t*hop + k - pad_left] * Bm[k, n];  x [R,L], Bm [W,N] -> [R*T, N]."""
    _chk(x, Bm)
    R, L = x.shape
    W, N = Bm.shape
    out = torch.empty((R * T, N), dtype=torch.float32, device=x.device)
    ev = PROFILE.begin() if PROFILE.enabled else None
    check(load().ams_frames_matmul(_p(x), _p(Bm), _p(out), R, L, W, N, hop, T, pad_left, _s()), 'ams_frames_matmul')
    if ev is not None:
        PROFILE.end(ev, 2.0 * R * T * N * W, 4.0 * (R * L + W * N + R * T * N), 'gemm<2,0>')
    return out


def overlap_add(frames, R, T, W, L, hop, pad_left):
    _chk(frames)
    out = torch.empty((R, L), dtype=torch.float32, device=frames.device)
    check(load().ams_overlap_add(_p(frames), _p(out), R, T, W, L, hop, pad_left, _s()), 'ams_overlap_add')
    return out


# ------------------------------------------------------------------ waveform statistics
def pair_stats_fwd(target, est, mix):
    _chk(target, est, mix)
    lib = load()
    B, S, L = est.shape
    NS = 2 * S * S + 3 * S + 1
    nb = lib.ams_pair_stats_workspace_bytes(B, S, L)
    ws = _ws(nb, est)
    stats = torch.empty((B, NS), dtype=torch.float32, device=est.device)
    check(lib.ams_pair_stats_fwd(_p(target), _p(est), _p(mix), _p(stats), B, S, L, _p(ws), nb, _s()), 'ams_pair_stats_fwd')
    return stats


def pair_stats_bwd(target, est, gstats):
    _chk(target, est, gstats)
    B, S, L = est.shape
    dest = torch.empty_like(est)
    check(load().ams_pair_stats_bwd(_p(target), _p(est), _p(gstats), _p(dest), B, S, L, _s()), 'ams_pair_stats_bwd')
    return dest


def pair_combine_fwd(stats, D2, perms, S, mode, cl, cs):
    """Costs from the pair table in one launch (csrc/synth.hip pair_combine_*): returns (out [2], pbest, jbest)."""
    _chk(stats)
    B = stats.shape[0]
    dev = stats.device
    out = torch.empty(2, dtype=torch.float32, device=dev)
    P = 0 if perms is None else perms.shape[0]
    pbest = torch.empty(B, dtype=torch.int32, device=dev) if mode != 0 else None
    jbest = torch.empty((B, S), dtype=torch.int32, device=dev) if mode == 2 else None
    check(load().ams_pair_combine_fwd(_p(stats), _p(D2), _p(perms), _p(out), _p(pbest), _p(jbest), B, S, P, mode, float(cl), float(cs),
                                      _s()), 'ams_pair_combine_fwd')
    return out, pbest, jbest


def pair_combine_bwd(stats, D2, perms, gout, pbest, jbest, S, mode, cl, cs):
    _chk(stats, gout)
    B = stats.shape[0]
    P = 0 if perms is None else perms.shape[0]
    gst = torch.empty_like(stats)
    gD2 = torch.empty_like(D2) if mode == 2 else None
    check(load().ams_pair_combine_bwd(_p(stats), _p(D2), _p(perms), _p(gout), _p(pbest), _p(jbest), _p(gst), _p(gD2), B, S, P, mode,
                                      float(cl), float(cs), _s()), 'ams_pair_combine_bwd')
    return gst, gD2


def apply_masks_fwd(X, masks):
    """X [B,TF], masks [B,TF,S] -> [B*S, TF]."""
    _chk(X, masks)
    B, TF, S = masks.shape
    sep = torch.empty((B * S, TF), dtype=torch.float32, device=X.device)
    check(load().ams_apply_masks_fwd(_p(X), _p(masks), _p(sep), B, S, TF, _s()), 'ams_apply_masks_fwd')
    return sep


def apply_masks_bwd(X, dsep, S):
    _chk(X, dsep)
    B, TF = X.shape
    dm = torch.empty((B, TF, S), dtype=torch.float32, device=X.device)
    check(load().ams_apply_masks_bwd(_p(X), _p(dsep), _p(dm), B, S, TF, _s()), 'ams_apply_masks_bwd')
    return dm


def overlap_metric_fwd(y, B, S):
    _chk(y)
    TN = y.numel() // (B * (S + 1))
    ws = _ws(4096, y)
    out = torch.empty(1, dtype=torch.float32, device=y.device)
    check(load().ams_overlap_metric_fwd(_p(y), _p(out), B, S, TN, _p(ws), 4096, _s()), 'ams_overlap_metric_fwd')
    return out


def overlap_metric_bwd(y, upstream, B, S):
    _chk(y, upstream)
    TN = y.numel() // (B * (S + 1))
    dy = torch.empty_like(y)
    check(load().ams_overlap_metric_bwd(_p(y), _p(upstream), _p(dy), B, S, TN, _s()), 'ams_overlap_metric_bwd')
    return dy


# ------------------------------------------------------------------ complex glue
def cplx_mag_phase(ri, F, want_phase=True):
    """ri [rows, >= 2F] = [Re | Im | padding] -> (|.| [rows, F], unit phasor [rows, 2F] or None)."""
    _chk(ri)
    rows = ri.shape[0]
    mag = torch.empty((rows, F), dtype=torch.float32, device=ri.device)
    ph = torch.empty((rows, 2 * F), dtype=torch.float32, device=ri.device) if want_phase else None
    check(load().ams_cplx_mag_phase(_p(ri), _p(mag), _p(ph), rows, F, ri.shape[1], _s()), 'ams_cplx_mag_phase')
    return mag, ph


def cplx_apply_fwd(sep, phasor, S, T):
    _chk(sep, phasor)
    rows, F = sep.shape
    z = torch.empty((rows, 2 * F), dtype=torch.float32, device=sep.device)
    check(load().ams_cplx_apply_fwd(_p(sep), _p(phasor), _p(z), rows, F, S, T, _s()), 'ams_cplx_apply_fwd')
    return z


def cplx_apply_bwd(dz, phasor, S, T):
    _chk(dz, phasor)
    rows, F2 = dz.shape
    d = torch.empty((rows, F2 // 2), dtype=torch.float32, device=dz.device)
    check(load().ams_cplx_apply_bwd(_p(dz), _p(phasor), _p(d), rows, F2 // 2, S, T, _s()), 'ams_cplx_apply_bwd')
    return d


# ------------------------------------------------------------------ L41 loss
def l41_loss_fwd(emb, y, vspk, from_u=False):
    _chk(emb, y, vspk)
    lib = load()
    B, TF, E = emb.shape
    S = y.shape[2]
    nb = lib.ams_l41_workspace_bytes(B, TF, E, S)
    ws = _ws(nb, emb)
    cost = torch.empty(1, dtype=torch.float32, device=emb.device)
    check(lib.ams_l41_loss_fwd(_p(emb), _p(y), _p(vspk), _p(cost), B, TF, E, S, int(from_u), _p(ws), nb, _s()), 'ams_l41_loss_fwd')
    return cost


def l41_loss_bwd(emb, y, vspk, upstream, from_u=False):
    _chk(emb, y, vspk, upstream)
    lib = load()
    B, TF, E = emb.shape
    S = y.shape[2]
    nb = lib.ams_l41_workspace_bytes(B, TF, E, S)
    ws = _ws(nb, emb)
    demb = torch.empty_like(emb)
    dvs = torch.empty_like(vspk)
    am = torch.empty(1, dtype=torch.float32, device=emb.device) if F16X3 else None      # max |demb| out of the same pass
    check(lib.ams_l41_loss_bwd(_p(emb), _p(y), _p(vspk), _p(upstream), _p(demb), _p(dvs), _p(am), B, TF, E, S, int(from_u), _p(ws), nb, _s()),
          'ams_l41_loss_bwd')
    tag_amax(demb, am)
    return demb, dvs


def l41_loss_ns_fwd(emb, y, vspk, negs, ns_rate, from_u=False):
    """L41 loss with negative sampling (L41.py:69-147,165-166): negs [B,NSEL,K,E], NSEL = 1 or S."""
    _chk(emb, y, vspk, negs)
    lib = load()
    B, TF, E = emb.shape
    S = y.shape[2]
    NSEL, K = negs.shape[1], negs.shape[2]
    nb = lib.ams_l41_ns_workspace_bytes(B, TF, E, S, NSEL, K)
    ws = _ws(nb, emb)
    cost = torch.empty(1, dtype=torch.float32, device=emb.device)
    check(lib.ams_l41_loss_ns_fwd(_p(emb), _p(y), _p(vspk), _p(negs), _p(cost), B, TF, E, S, NSEL, K, float(ns_rate), int(from_u), _p(ws), nb,
                                  _s()),
          'ams_l41_loss_ns_fwd')
    return cost


def l41_loss_ns_bwd(emb, y, vspk, negs, upstream, ns_rate, from_u=False):
    _chk(emb, y, vspk, negs, upstream)
    lib = load()
    B, TF, E = emb.shape
    S = y.shape[2]
    NSEL, K = negs.shape[1], negs.shape[2]
    nb = lib.ams_l41_ns_workspace_bytes(B, TF, E, S, NSEL, K)
    ws = _ws(nb, emb)
    demb, dvs, dnegs = torch.empty_like(emb), torch.empty_like(vspk), torch.empty_like(negs)
    am = torch.empty(1, dtype=torch.float32, device=emb.device) if F16X3 else None
    check(lib.ams_l41_loss_ns_bwd(_p(emb), _p(y), _p(vspk), _p(negs), _p(upstream), _p(demb), _p(dvs), _p(dnegs), _p(am), B, TF, E, S, NSEL, K,
                                  float(ns_rate), int(from_u), _p(ws), nb, _s()), 'ams_l41_loss_ns_bwd')
    tag_amax(demb, am)
    return demb, dvs, dnegs


# ------------------------------------------------------------------ k-means
def kmeans_normalize(x):
    _chk(x)
    E = x.shape[-1]
    xn = torch.empty_like(x)
    check(load().ams_kmeans_normalize(_p(x), _p(xn), x.numel() // E, E, _s()), 'ams_kmeans_normalize')
    return xn


_KM_TICKETS = {}


def kmeans_run(xn, init_idx, C, tries, iterations, beta=None, w=None, assign_at_end=True, faithful_tile=True):
    """Whole KMeans.network (Kmeans_2.py:86-111) on normalised input xn [b,L,E].
    init_idx int32 [b*tries, C].  Returns (centroids [b,C,E], labels int32 [b,L] | soft [b,L,C], best [b], cent_trace)
    where cent_trace is the list of per-iteration centroid tensors [R,C,E] (needed by the soft backward)."""
    _chk(xn, w)
    lib = load()
    b, L, E = xn.shape
    R = b * tries
    dev = xn.device
    if tuple(init_idx.shape) != (R, C) or init_idx.dtype != torch.int32 or not init_idx.is_cuda:
        raise AmsError('kmeans: init_idx must be an int32 device tensor of shape [b*tries, C] = %s, got %s %s'
                       % ((R, C), tuple(init_idx.shape), init_idx.dtype))
    hard = beta is None
    bval = -1.0 if hard else float(beta)
    nb = lib.ams_kmeans_workspace_bytes(R, L, E, C)
    ws = _ws(nb, xn)
    # row tickets of the in-launch finish (include/ams.h): persistent per device and stream, zero once, every pass leaves them zero
    tk = _KM_TICKETS.get((dev.index, torch.cuda.current_stream().cuda_stream))
    if tk is None or tk.numel() < R:
        tk = _KM_TICKETS[(dev.index, torch.cuda.current_stream().cuda_stream)] = torch.zeros(max(R, 1024), dtype=torch.int32, device=dev)
    # soft: the centroids and denominators of all iterations are slices of TWO stacked buffers -- what the backward pass wants in one
    # piece (kmeans_bwd.soft_bwd: no torch.stack of 21 small tensors in front of its first launch)
    cstack = torch.empty((iterations + 1, R, C, E), dtype=torch.float32, device=dev) if not hard else None
    dstack = torch.empty((max(iterations, 1), R, C), dtype=torch.float32, device=dev) if not hard else None
    cent = cstack[0] if cstack is not None else torch.empty((R, C, E), dtype=torch.float32, device=dev)
    check(lib.ams_kmeans_init(_p(xn), _p(init_idx), _p(cent), b, tries, L, E, C, _s()), 'ams_kmeans_init')
    trace = [cent]
    wm = 1 if faithful_tile else 0
    for it in range(iterations):
        nxt = cstack[it + 1] if cstack is not None else torch.empty_like(cent)
        den = dstack[it] if dstack is not None else None
        check(lib.ams_kmeans_iterate(_p(xn), _p(w), _p(cent), _p(nxt), _p(den), b, tries, L, E, C, bval, wm, _p(ws), nb, _p(tk), _s()),
              'ams_kmeans_iterate')
        cent = nxt
        trace.append(cent)
        if den is not None:
            trace.append(den)           # soft: trace = [c_0, c_1, den_0, c_2, den_1, ...]
    inertia = torch.empty(R, dtype=torch.float32, device=dev)
    labels = torch.empty((R, L), dtype=torch.int32, device=dev) if (hard and not assign_at_end) else None
    soft = torch.empty((R, L, C), dtype=torch.float32, device=dev) if (not hard and not assign_at_end) else None
    check(lib.ams_kmeans_assign(_p(xn), _p(w), _p(cent), _p(labels), _p(soft), _p(inertia), b, tries, L, E, C, bval, wm, _p(ws), nb,
                                _p(tk), _s()), 'ams_kmeans_assign')
    best = torch.empty(b, dtype=torch.int32, device=dev)
    sel = torch.empty((b, C, E), dtype=torch.float32, device=dev)
    check(lib.ams_kmeans_select(_p(inertia), _p(cent), _p(best), _p(sel), b, tries, E, C, _s()), 'ams_kmeans_select')
    if assign_at_end:
        out_l = torch.empty((b, L), dtype=torch.int32, device=dev) if hard else None
        out_s = torch.empty((b, L, C), dtype=torch.float32, device=dev) if not hard else None
        check(lib.ams_kmeans_assign(_p(xn), _p(None), _p(sel), _p(out_l), _p(out_s), _p(None), b, 1, L, E, C, bval, 0, _p(ws), nb, _p(None), _s()),
              'ams_kmeans_assign(end)')
        out = out_l if hard else out_s
    else:
        idx = (best.long() + torch.arange(b, device=dev) * tries)
        out = (labels if hard else soft)[idx]
    return sel, out, best, trace
