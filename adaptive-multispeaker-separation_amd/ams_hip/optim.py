"""Flat-buffer optimizers and gradient exchange (reference models/network.py:167-194, utils/ops.py:639-792).

All trainable variables are re-pointed into ONE contiguous fp32 buffer; their gradients live in a second
buffer of the same layout.  That buffer is what RCCL all-reduces (one collective per step over xGMI) and
what the fused optimizer kernel walks -- one launch instead of one tiny kernel per variable per moment.
"""
import math

import torch

from . import ops


class FlatOptimizer(object):
    def __init__(self, variables, kind, learning_rate, decay_epoch, gradient_clip, dist=None):
        self.vars = list(variables)
        self.kind = kind
        self.base_lr = float(learning_rate)
        self.decay_epoch = int(decay_epoch)
        self.clip = float(gradient_clip)
        self.dist = dist
        self.global_epoch = 0
        self.t = 0
        n = sum(v.numel() for v in self.vars)
        dev = self.vars[0].device if self.vars else torch.device('cpu')
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        # Data parallel on GPUs: ONE extra element behind the gradients carries the recurrence rings' error word through the SAME
        # all-reduce (a rank whose ring gave up must make every rank skip the update and repeat the step): one collective per step
        # instead of an all_reduce_max of the word followed by the all_reduce of the gradients.
        self._dp = dist is not None and getattr(dist, 'world_size', 1) > 1
        extra = 1 if (self._dp and dev.type == 'cuda') else 0
        # (padded to whole 16-byte groups: cleared by the fused zero launch at the head of a step, ops.pass_begin)
        self._gbuf = torch.zeros((n + extra + 3) // 4 * 4, dtype=torch.float32, device=dev)
        self.flat_grad = self._gbuf[:n]
        self._errslot = self._gbuf[n:n + 1] if extra else None
        if self._errslot is not None:
            ops.register_error_word(self._errslot)
        self.exchange_events = None              # bench.py: list of (start, end) events around the gradient all-reduce
        # AMS_DP_OVERLAP=1 (off by default: never run on multi-GPU hardware yet): gradients are exchanged in BUCKETS as the backward
        # pass finishes them -- bucket_ready() all-reduces the flat-buffer range of a layer on a communication stream behind the side
        # stream that wrote it; exchange() then only reduces what is left (the first layer, the error word) and joins.  Eager steps
        # only: a captured step keeps the single exchange (bucket_ready warns once).
        import os
        self.overlap = self._dp and os.environ.get('AMS_DP_OVERLAP', '0') == '1'
        self._done = []                          # [lo, hi) element ranges of _gbuf already all-reduced in this step
        self._comm = None
        off = 0
        ids = set(id(v) for v in self.vars)
        placed = set()
        for v in self.vars:
            if id(v) in placed:
                continue
            k = v.numel()
            w = getattr(v, '_ams_twin', None)
            if w is not None and id(w) in ids and id(w) not in placed and w.shape == v.shape:
                # Twin variables (the two directions of a BLSTM layer) are stored ROW-INTERLEAVED: block [R, 2, C] with
                # v = block[:, 0, :], w = block[:, 1, :].  [Wx_f | Wx_b] is then a plain [D, 8H] row-major matrix, so
                # the merged-direction GEMMs read weights and write weight gradients in place (no gather/scatter copies).
                R = v.shape[0] if v.dim() == 2 else 1
                C = v.shape[-1]
                for buf, init in ((self.flat, True), (self.flat_grad, False)):
                    blk = buf[off:off + 2 * k].view(R, 2, C)
                    for j, t in enumerate((v, w)):
                        view = blk[:, j, :] if t.dim() == 2 else blk[0, j, :]
                        if init:
                            view.copy_(t.detach())
                            t.data = view
                            t.requires_grad_(True)
                        else:
                            t.grad = view
                placed.update((id(v), id(w)))
                off += 2 * k
                continue
            self.flat[off:off + k].copy_(v.detach().reshape(-1))
            v.data = self.flat[off:off + k].view(v.shape)
            v.requires_grad_(True)
            v.grad = self.flat_grad[off:off + k].view(v.shape)
            placed.add(id(v))
            off += k
        z = lambda: torch.zeros(n, dtype=torch.float32, device=dev)  # noqa: E731
        if kind == 'Adam':                      # the reference's 'Adam' is AMSGrad(eps=1e-3, beta2=.99), undecayed lr
            self.m, self.v, self.vhat = z(), z(), z()
            self.beta1, self.beta2, self.eps = 0.9, 0.99, 1e-3
            self.b1p, self.b2p = self.beta1, self.beta2
        elif kind == 'RMSProp':
            self.ms = torch.ones(n, dtype=torch.float32, device=dev)
        elif kind == 'SGD':
            self.acc = z()
        else:
            raise ValueError('unknown optimizer %r' % kind)
        # fp16x3 products (ops.set_amax) scale their weight operand by ONE bound over everything this optimizer owns, measured
        # at its first use in every pass (ops.param_amax)
        if ops.F16X3 and dev.type == 'cuda' and n > 0:
            ops.register_param_source(self.vars, self.flat)

    # tf.train.exponential_decay(lr, global_epoch, decay_epoch, 0.5, staircase=True)  (network.py:175-177)
    def learning_rate(self):
        return self.base_lr * 0.5 ** (self.global_epoch // self.decay_epoch)

    def increment_epoch(self):
        self.global_epoch += 1

    def zero_grad(self, defer=False):
        """defer: zeroed on the side stream when the next evaluation pass begins (ops.zero_deferred) -- for callers whose backward
        pass starts with ops.flush_deferred_zero() + ops.await_pass_side() (Network._backward)."""
        if defer:
            ops.zero_deferred(self._gbuf)
        else:
            self._gbuf.zero_()

    def range_of(self, *params):
        """[lo, hi) element range of the flat gradient buffer spanned by the gradients of `params` (a layer's variables are adjacent;
        a twin-interleaved variable's rows alternate with its partner's, so the partner is part of the span)."""
        base, es = self._gbuf.data_ptr(), self._gbuf.element_size()
        lo, hi = None, None
        for p in params:
            for t in (p, getattr(p, '_ams_twin', None)):
                if t is None or t.grad is None:
                    continue
                g = t.grad
                start = (g.data_ptr() - base) // es
                span = (g.shape[0] - 1) * g.stride(0) + g.shape[1] if g.dim() == 2 else g.numel()
                lo = start if lo is None else min(lo, start)
                hi = start + span if hi is None else max(hi, start + span)
        return int(lo), int(hi)

    def bucket_ready(self, params, after_stream=None):
        """The gradients of `params` are final (written by work already enqueued on `after_stream`): start their all-reduce now."""
        if not self.overlap or any(p.grad is None for p in params):
            return
        gb = self._gbuf
        if gb.is_cuda and torch.cuda.is_current_stream_capturing():
            # A bucket issued while the step is being CAPTURED would (a) fork the communication stream from the capturing stream and leave
            # it unjoined when the capture ends -- exchange(), which joins it, runs outside -- and (b) fill _done at capture time only, so
            # that every replay all-reduced the bucket inside the graph AND again, whole, in exchange().  The captured step
            # therefore keeps the single exchange, whatever the backend, and says so once.
            if not getattr(self, '_warned_capture', False):
                import warnings
                warnings.warn('AMS_DP_OVERLAP=1 has no effect inside a captured step (--hip_graph): the gradients are exchanged by ONE '
                              'all-reduce after the replay', RuntimeWarning)
                self._warned_capture = True
            return
        lo, hi = self.range_of(*params)
        # ranges never overlap: a layer's variables are adjacent in the flat buffer and every layer announces itself once per step.  A
        # gradient accumulated into a range AFTER its all-reduce was issued would never be reduced -- refuse that loudly instead
        if any(not (hi <= a or lo >= b) for a, b in self._done):
            raise RuntimeError('bucket_ready: range [%d, %d) overlaps a bucket already exchanged in this step %r (a variable shared '
                               'between layers cannot be bucketed)' % (lo, hi, self._done))
        if gb.is_cuda:
            if self._comm is None:
                self._comm = torch.cuda.Stream()
            self._comm.wait_stream(after_stream if after_stream is not None else torch.cuda.current_stream())
            with torch.cuda.stream(self._comm):
                self.dist.all_reduce_sum(gb[lo:hi])
        else:
            self.dist.all_reduce_sum(gb[lo:hi])
        self._done.append((lo, hi))

    def exchange(self):
        """Data-parallel gradient exchange: ONE all-reduce (sum) of the flat gradient buffer; returns the scale that
        turns it into the mean (every loss on the hot path is a batch mean) and applies the clip."""
        scale = 1.0
        if self._dp:
            if self._errslot is not None and ops.LSTM_RING != '0':
                self._errslot.copy_(ops.ring_error_word(self.flat.device))     # int32 0 / 1 -> float: summed over the ranks with the gradients
            ev = None
            if self.exchange_events is not None and self._gbuf.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self._done:
                # what bucket_ready() has not sent: the gaps between its ranges (first layer, padding, the error word)
                pos = 0
                for a, b in sorted(self._done) + [(self._gbuf.numel(), self._gbuf.numel())]:
                    if a > pos:
                        self.dist.all_reduce_sum(self._gbuf[pos:a])
                    pos = max(pos, b)
                if self._comm is not None:
                    torch.cuda.current_stream().wait_stream(self._comm)
                self._done = []
            else:
                self.dist.all_reduce_sum(self._gbuf)
            if ev is not None:
                ev[1].record()
                self.exchange_events.append(ev)
            scale = 1.0 / self.dist.world_size
        self._scale_dev = None
        if self.clip != 0.0:
            if self.flat_grad.is_cuda:
                # norm of the AVERAGED gradient and tf.clip_by_global_norm's factor stay on the device (no .item() between the
                # all-reduce and the update): the optimizer kernel multiplies it in (include/ams.h: grad_scale_dev)
                self._scale_dev = ops.clip_scale(ops.sumsq(self.flat_grad), scale, self.clip)
            else:
                gn = math.sqrt(float(self._sumsq(self.flat_grad))) * scale
                scale *= self.clip / max(gn, self.clip)
        return scale

    def undo_counters(self):
        """Host-side step counters back to before the last step() (whose device update the guard word skipped)."""
        b1p, b2p, t = getattr(self, '_before', (None, None, self.t))
        if b1p is not None:
            self.b1p, self.b2p = b1p, b2p
        self.t = t

    def _sumsq(self, t):
        return ops.sumsq(t).item() if t.is_cuda else float((t.double() ** 2).sum())

    def step(self):
        """Gradient exchange + clip + fused update.  Gradients must already be in flat_grad."""
        if self.flat.numel() == 0:
            return
        self._before = (getattr(self, 'b1p', None), getattr(self, 'b2p', None), self.t)
        scale = self.exchange()
        # a rank whose recurrence ring gave up makes EVERY rank skip this update and repeat the step (the gradients it contributed to
        # the all-reduce are garbage): under data parallelism the guard is the summed word that travelled with the gradients
        guard = self._errslot if (self._errslot is not None and ops.LSTM_RING != '0') else None
        if self.kind == 'Adam':
            lr_t = self.base_lr * math.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
            ops.opt_amsgrad(self.flat, self.flat_grad, self.m, self.v, self.vhat, lr_t, self.beta1, self.beta2, self.eps, scale, guard=guard, scale_dev=self._scale_dev)
            self.b1p *= self.beta1
            self.b2p *= self.beta2
        elif self.kind == 'RMSProp':
            ops.opt_rmsprop(self.flat, self.flat_grad, self.ms, self.learning_rate(), 0.9, 1e-10, scale, guard=guard, scale_dev=self._scale_dev)
        else:
            ops.opt_momentum(self.flat, self.flat_grad, self.acc, self.learning_rate(), 0.9, scale, guard=guard, scale_dev=self._scale_dev)
        self.t += 1
