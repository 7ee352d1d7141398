"""Host side of the batched k-means (reference models/Kmeans_2.py:12-195): same constructor and `.network` /
`.fit` surface; the Lloyd iterations, restarts, inertia selection and final assignment run in csrc/kmeans.hip."""
import numpy as np
import torch

from . import functional as F
from .graph import Node, Placeholder, get_default_graph


class KMeans(object):

    def __init__(self, nb_clusters, centroids_init=None, nb_tries=10, nb_iterations=10, input_tensor=None,
                 normalize_input=True, latent_space_tensor=None, beta=None, threshold=2.5, assign_at_end=True,
                 init_indices=None, seeding='reference'):
        if seeding not in ('reference', 'fast'):
            raise ValueError("seeding must be 'reference' or 'fast', got %r" % (seeding,))
        self.seeding = seeding
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
        (a full Fisher-Yates shuffle of l bins per row: ~0.15 ms x 640 rows at the benchmark shape, far above the 6 ms the GPU
        needs for the batch), so throughput runs may opt into `seeding='fast'` (--kmeans_seeding fast): the same distribution
        (C distinct bins, uniform) in one vectorised call from the same global RNG, rows with a repeated index redrawn -- a
        DIFFERENT stream, hence different (equally valid) restarts."""
        C = self.nb_clusters
        if self.seeding == 'reference':
            a = np.array([np.random.choice(L, size=C, replace=False) for _ in range(R)])
            return torch.from_numpy(a.astype(np.int32))
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
        X = self.X_in.value(run)
        b, L, E = X.shape
        w = None
        if self.latent_space_tensor is not None:
            from . import ops
            lat = self.latent_space_tensor.value(run).reshape(b, L).contiguous()
            w = ops.silence_weights(lat, self.threshold)                                        # Kmeans_2.py:76-80
        idx = self._init_idx(run, b * self.nb_tries, L, X.device)
        return F.kmeans(X, idx, self.nb_clusters, self.nb_tries, self.nb_iterations, self.beta, w, self.assign_at_end,
                        self.normalize_input, self.faithful_tile)

    def fit(self, X_train):
        from .graph import Run
        x = torch.as_tensor(X_train, dtype=torch.float32, device=get_default_graph().device)
        run = Run({self.X_in: x}, training=False)
        with torch.no_grad():
            c, l = self._result.value(run)[:2]
        return c.cpu().numpy(), l.cpu().numpy()
