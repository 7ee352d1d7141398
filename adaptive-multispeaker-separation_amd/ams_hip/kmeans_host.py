"""Host side of the batched k-means (reference models/Kmeans_2.py:12-195): same constructor and `.network` /
`.fit` surface; the Lloyd iterations, restarts, inertia selection and final assignment run in csrc/kmeans.hip."""
import os
import threading

import numpy as np
import torch

from . import functional as F
from .graph import Node, Placeholder, get_default_graph


class _ReferenceSeeds(object):
    """np.random.choice(range(l), size=C, replace=False) for R rows from numpy's GLOBAL generator (Kmeans_2.py:61-66), through
    libams_host.so::ams_mt_choice_rows (csrc/host/mt_choice.c): the same values and the same stream position afterwards, ~15x
    faster than the per-row numpy calls (which shuffle all l bins to keep C of them).

    One batch ahead: after serving a draw the worker thread computes the NEXT draw of the same shape from the state this one left
    (ctypes releases the GIL, so it runs beside the GPU launches of the current batch).  The speculation is used only if, at the
    next call, numpy's global state is still exactly the state it started from (nobody else drew in between) and the shape is
    the same; otherwise it is dropped and the draw is made on the spot -- the stream numpy users see never depends on it."""

    def __init__(self):
        self._fn = None
        self._spec = None               # (thread, key0, pos0, (R, L, C), box)
        self.ahead = os.environ.get('AMS_KMEANS_SEED_AHEAD', '1') != '0'
        self.hits = 0

    def _native(self):
        if self._fn is None:
            import ctypes
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libams_host.so')
            try:
                fn = ctypes.CDLL(path).ams_mt_choice_rows
                fn.restype = ctypes.c_int
                fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
                self._fn = fn
            except (OSError, AttributeError):
                self._fn = False
        return self._fn

    def _run(self, key, pos, shape, box):
        R, L, C = shape
        out = np.empty((R, C), np.int32)
        p = np.array([pos], np.int32)
        rc = self._fn(key.ctypes.data, p.ctypes.data, R, L, C, out.ctypes.data)
        box.append((rc, out, key, int(p[0])))

    def draw(self, R, L, C):
        fn = self._native()
        st = np.random.get_state()
        if not fn or st[0] != 'MT19937' or C > L or R == 0:          # numpy's own path (and its own error for C > L)
            return np.array([np.random.choice(L, size=C, replace=False) for _ in range(R)]).astype(np.int32).reshape(R, C)
        key0, pos0, shape = np.ascontiguousarray(st[1], dtype=np.uint32), int(st[2]), (R, L, C)
        box, spec, self._spec = None, self._spec, None
        if spec is not None:
            th, k, p, sh, b = spec
            th.join()
            if sh == shape and p == pos0 and np.array_equal(k, key0):
                box = b
                self.hits += 1
        if box is None:
            box = []
            self._run(key0.copy(), pos0, shape, box)
        rc, out, key1, pos1 = box[0]
        if rc != 0:
            raise RuntimeError('ams_mt_choice_rows failed (%d)' % rc)
        np.random.set_state((st[0], key1, pos1, st[3], st[4]))
        if self.ahead:
            nb = []
            th = threading.Thread(target=self._run, args=(key1.copy(), pos1, shape, nb), daemon=True)
            th.start()
            self._spec = (th, key1.copy(), pos1, shape, nb)
        return out


_REFERENCE_SEEDS = _ReferenceSeeds()


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def keyed_seeds(draw, row0, R, L, C, seed=42):
    """int32 [R, C]: C distinct bins in [0, L) for the GLOBAL rows row0 .. row0 + R - 1 of draw number `draw`: a counter-based stream
    (splitmix64 of (seed, draw, global row, cluster, attempt)), so a row's seeds depend on nothing but its global index -- the same
    whatever the number of ranks and their shard sizes.  Same distribution as the reference's np.random.choice(l, C, replace=False)
    (uniform C-subsets in uniform order: rows with a repeated bin are redrawn), a different stream."""
    rows = (np.arange(R, dtype=np.uint64) + np.uint64(row0))[:, None]
    cl = np.arange(C, dtype=np.uint64)[None, :]
    out = np.zeros((R, C), dtype=np.int64)
    todo = np.arange(R)
    attempt = 0
    with np.errstate(over='ignore'):
        base = _mix64(np.full((1, 1), seed, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(draw))
        while todo.size:
            h = _mix64(_mix64(base ^ (rows[todo] * np.uint64(0xD1342543DE82EF95))) + cl * np.uint64(0x2545F4914F6CDD1D)
                       + np.uint64(attempt) * np.uint64(0x9E3779B97F4A7C15))
            a = ((h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53)) * L).astype(np.int64)      # uniform in [0, L)
            out[todo] = a
            srt = np.sort(a, axis=1)
            todo = todo[(srt[:, 1:] == srt[:, :-1]).any(axis=1)] if C > 1 else todo[:0]
            attempt += 1
    return out.astype(np.int32)


class KMeans(object):

    def __init__(self, nb_clusters, centroids_init=None, nb_tries=10, nb_iterations=10, input_tensor=None,
                 normalize_input=True, latent_space_tensor=None, beta=None, threshold=2.5, assign_at_end=True,
                 init_indices=None, seeding='reference', pre_norm=None, dist=None):
        if seeding not in ('reference', 'fast', 'keyed'):
            raise ValueError("seeding must be 'reference', 'fast' or 'keyed', got %r" % (seeding,))
        # Data parallel: the host streams of 'reference' / 'fast' are consumed row by row, so with G ranks utterance j of EVERY shard
        # would get rank 0's seeds and results would depend on G.  Ranks > 1 therefore draw 'keyed' seeds -- a counter-based stream
        # indexed by (seed 42, draw number, GLOBAL row) (SURVEY 8e "Partitioning") -- unless the caller asked for something else by name.
        self.dist = dist
        if dist is not None and getattr(dist, 'enabled', False) and seeding == 'reference':
            seeding = 'keyed'
        self.seeding = seeding
        self._draws = 0
        if centroids_init is not None:
            raise NotImplementedError('explicit centroids_init (Kmeans_2.py:72-74) is unused by the reference recipes')
        self.nb_clusters = nb_clusters
        self.nb_iterations = nb_iterations
        self.nb_tries = nb_tries
        self.latent_space_tensor = latent_space_tensor
        self.beta = beta
        self.assign_at_end = assign_at_end
        self.threshold = threshold
        self.normalize_input = normalize_input
        self.init_indices = init_indices          # Node / callable giving int32 [b*tries, C]; None -> host RNG as the reference
        self.faithful_tile = True
        # (node, E) or None: input_tensor is l2-normalise(node) over groups of E -- a soft k-means under gradient that normalises its input
        # anyway then starts from that node (functional.KMeansSoft, from_u) and input_tensor is not evaluated
        self.pre_norm = pre_norm
        g = get_default_graph()
        with g.variable_scope('kmeans'):
            self.X_in = input_tensor if input_tensor is not None else Placeholder('Kmeans_input')
            self._result = Node('network', self._run)
        self.network = (Node('centroids', lambda run: self._result.value(run)[0], register=False),
                        Node('labels', lambda run: self._result.value(run)[1], register=False))

    def _draw(self, R, L):
        """Seed bins, int32 [R, C].  `seeding='reference'` (default) IS Kmeans_2.py:61-66: one
        np.random.choice(range(l), size=C, replace=False) per row, in row order, from the GLOBAL numpy RNG (seeded 42 at
        models/network.py:17-18) -- index work is bit-exact with the reference given the same stream position
        (tests/test_host_mirror.py::test_kmeans_reference_seeding).  `choice(l, ...)` consumes exactly the stream of
        `choice(range(l), ...)` (both are permutation(l)[:C]) without building a list per row.  The draw is inherently serial
        (every row consumes a data-dependent number of MT19937 words: ~27 k for l = 20480); _ReferenceSeeds makes it through the
        C helper, one batch ahead on a worker thread, which still costs ~20-40 us x 640 rows at the benchmark shape -- above the
        6 ms the GPU needs for the batch -- so throughput runs may opt into `seeding='fast'` (--kmeans_seeding fast): the same distribution
        (C distinct bins, uniform) in one vectorised call from the same global RNG, rows with a repeated index redrawn -- a
        DIFFERENT stream, hence different (equally valid) restarts."""
        C = self.nb_clusters
        if self.seeding == 'keyed':
            d = self.dist
            world = d.world_size if (d is not None and getattr(d, 'enabled', False)) else 1
            row0 = 0
            if world > 1:
                # GLOBAL row of this rank's first row = rows held by the ranks before it IN THIS DRAW (a ragged last batch gives the
                # ranks different R: rank * R would overlap or skip rows), and every rank must be at the same draw number -- both
                # from one small host gather per draw (hard k-means under data parallelism only)
                mine = (int(R), int(self._draws))
                every = d.all_gather_object(mine)
                if any(e[1] != mine[1] for e in every):
                    raise RuntimeError('keyed k-means seeds: ranks are at different draw numbers %r -- a rank skipped a draw' % (every,))
                row0 = sum(e[0] for e in every[:d.rank])
            out = keyed_seeds(self._draws, row0, R, L, C)
            self._draws += 1
            return torch.from_numpy(out)
        if self.seeding == 'reference':
            return torch.from_numpy(_REFERENCE_SEEDS.draw(R, L, C))
        a = np.random.randint(0, L, size=(R, C))
        if C > 1:
            while True:
                srt = np.sort(a, axis=1)
                bad = np.flatnonzero((srt[:, 1:] == srt[:, :-1]).any(axis=1))
                if bad.size == 0:
                    break
                a[bad] = np.random.randint(0, L, size=(bad.size, C))
        return torch.from_numpy(a.astype(np.int32))

    def _init_idx(self, run, R, L, device):
        if self.init_indices is not None:
            v = self.init_indices.value(run) if hasattr(self.init_indices, 'value') else self.init_indices
            return torch.as_tensor(np.asarray(v), dtype=torch.int32, device=device).contiguous()
        # persistent device buffer: under hipGraph capture the host draw cannot be part of the graph, so the buffer is
        # re-filled by a pre-replay hook (graph.pre_replay_hooks) and the captured kernels just read it
        buf = getattr(self, '_idx_buf', None)
        capturing = device.type == 'cuda' and torch.cuda.is_current_stream_capturing()
        if buf is None or tuple(buf.shape) != (R, self.nb_clusters) or buf.device != device:
            if capturing:
                raise RuntimeError('k-means seed buffer must exist before hipGraph capture (run an eager step first)')
            buf = self._idx_buf = torch.empty((R, self.nb_clusters), dtype=torch.int32, device=device)
            self._idx_host = torch.empty((R, self.nb_clusters), dtype=torch.int32)
            if device.type == 'cuda':
                self._idx_host = self._idx_host.pin_memory()        # allocated eagerly: pinning is not allowed while capturing
            self._hooked = False
        if capturing:
            if not getattr(self, '_hooked', False):
                host = self._idx_host

                def _refresh(buf=buf, host=host, R=R, L=L):
                    host.copy_(self._draw(R, L))
                    buf.copy_(host, non_blocking=True)
                get_default_graph().pre_replay_hooks.append(_refresh)
                self._hooked = True
        else:
            buf.copy_(self._draw(R, L))
        return buf

    def _run(self, run):
        pre = None
        soft_grad = self.beta is not None and run.training and torch.is_grad_enabled()
        no_grad = self.beta is None or not torch.is_grad_enabled()
        if (self.pre_norm is not None and self.normalize_input and (soft_grad or no_grad) and id(self.X_in) not in run.cache):
            u = self.pre_norm[0].value(run)
            if u.is_cuda and (u.requires_grad or no_grad):
                E = self.pre_norm[1]
                b = u.shape[0]
                L = u.numel() // (b * E)
                pre = lambda: u.reshape(b, L, E)                                               # noqa: E731
                X, dev = (lambda: self.X_in.value(run)), u.device
        if pre is None:
            X = self.X_in.value(run)
            b, L, E = X.shape
            dev = X.device
        w = None
        if self.latent_space_tensor is not None:
            from . import ops
            lat = self.latent_space_tensor.value(run).reshape(b, L).contiguous()
            w = ops.silence_weights(lat, self.threshold)                                        # Kmeans_2.py:76-80
        idx = self._init_idx(run, b * self.nb_tries, L, dev)
        return F.kmeans(X, idx, self.nb_clusters, self.nb_tries, self.nb_iterations, self.beta, w, self.assign_at_end,
                        self.normalize_input, self.faithful_tile, pre_norm=pre)

    def fit(self, X_train):
        from .graph import Run
        x = torch.as_tensor(X_train, dtype=torch.float32, device=get_default_graph().device)
        run = Run({self.X_in: x}, training=False)
        with torch.no_grad():
            c, l = self._result.value(run)[:2]
        return c.cpu().numpy(), l.cpu().numpy()
