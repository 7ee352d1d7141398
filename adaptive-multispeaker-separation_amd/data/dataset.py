"""Input pipeline mirror (reference data/dataset.py:532-645 ``TFDataset``) backed by a synthetic-mixture source.

The reference builds a tf.data pipeline over LibriSpeech TFRecords (decode -> chunk -> zip S speakers -> sum);
those need TF/h5py/librosa and real audio and are out of scope for the GPU hot path.
This class honours the same OUTPUT contract -- ``next_mix [B,L]``, ``next_non_mix [B,S,L]``, ``next_ind [B,S]``
with ``x_mix = sum_s x_s`` exactly as dataset.py:462-468 -- from the synthetic generator SURVEY 8(d) specifies.
"""
import numpy as np
import torch

from ams_hip.graph import Node, Placeholder, get_default_graph

FS = 8000


def synthetic_speaker(i, s, L):
    """Harmonic-plus-noise 'speaker' s of mixture i: seed = 42 + 1000*i + s."""
    rng = np.random.RandomState((42 + 1000 * i + s) % (2 ** 31 - 1))
    f0 = rng.uniform(90.0, 250.0)
    t = np.arange(L, dtype=np.float64) / FS
    k = np.arange(1, 21, dtype=np.float64)
    a = (1.0 / k) * rng.uniform(0.5, 1.0, 20)
    phi = rng.uniform(0.0, 2.0 * np.pi, 20)
    keep = k * f0 < FS / 2.0
    x = (a[keep, None] * np.sin(2.0 * np.pi * k[keep, None] * f0 * t[None, :] + phi[keep, None])).sum(axis=0)
    # 4-6 Hz raised-cosine on/off syllable gate with random phase
    fr = rng.uniform(4.0, 6.0)
    ph = rng.uniform(0.0, 2.0 * np.pi)
    env = np.clip(0.5 - 0.5 * np.cos(2.0 * np.pi * fr * t + ph), 0.0, 1.0)
    env = np.where(np.cos(2.0 * np.pi * 0.5 * fr * t + ph) > -0.3, env, 0.0)
    x = env * x + 0.005 * rng.standard_normal(L)
    x *= 0.05 / max(np.sqrt(np.mean(x * x)), 1e-12)
    return x.astype(np.float32)


def synthetic_mixtures(indices, S, L, tot_speakers=251):
    """Returns (mix [n,L], non_mix [n,S,L], ind [n,S] int32) for global utterance indices."""
    n = len(indices)
    non_mix = np.zeros((n, S, L), dtype=np.float32)
    ind = np.zeros((n, S), dtype=np.int32)
    for j, i in enumerate(indices):
        for s in range(S):
            non_mix[j, s] = synthetic_speaker(int(i), s, L)
        rng = np.random.RandomState((7 + 31 * int(i)) % (2 ** 31 - 1))
        ind[j] = rng.choice(tot_speakers, size=S, replace=False)       # distinct speakers (dataset.py:473-480)
    mix = non_mix.sum(axis=1)                                           # dataset.py:462-468
    return mix, non_mix, ind


class TFDataset(object):
    TRAIN, VALID, TEST, TEST_OTHER = 'train', 'valid', 'test', 'test_other'

    def __init__(self, **kwargs):
        self.batch_size = kwargs['batch_size']
        self.S = kwargs['nb_speakers']
        self.default_chunk = kwargs['chunk_size']
        self.dist = kwargs.get('dist')
        self.device = get_default_graph().device
        nb = kwargs.get('synthetic_batches') or 20
        self.nb_batches = {self.TRAIN: nb, self.VALID: max(1, nb // 10), self.TEST: max(1, nb // 10),
                           self.TEST_OTHER: max(1, nb // 10)}
        self.offsets = {self.TRAIN: 0, self.VALID: 1000000, self.TEST: 2000000, self.TEST_OTHER: 3000000}
        self.cursor = dict.fromkeys(self.nb_batches, 0)
        self.pool = {}
        self.pool_batches = int(kwargs.get('synthetic_pool') or 8)

        g = get_default_graph()
        with g.variable_scope('dataset'):
            self.handle = Placeholder('handle')
            self.chunk_size = Placeholder('chunk_size')
            batch = Node('batch', self._next)
            self.next_mix = Node('next_mix', lambda run: batch.value(run)[0])
            self.next_non_mix = Node('next_non_mix', lambda run: batch.value(run)[1])
            self.next_ind = Node('next_ind', lambda run: batch.value(run)[2])
        self.training_initializer = self.TRAIN
        self.validation_initializer = self.VALID
        self.test_initializer = self.TEST

    # -- reference API ---------------------------------------------------------------
    def get_handle(self, split):
        return split

    def get_initializer(self, split):
        return split

    def initialize(self, split):
        self.cursor[split] = 0

    def length(self, split):
        return self.nb_batches[split]

    # -- batches -----------------------------------------------------------------------
    def _batch_indices(self, split, b):
        world = self.dist.world_size if self.dist is not None else 1
        rank = self.dist.rank if self.dist is not None else 0
        base = self.offsets[split] + (b * world + rank) * self.batch_size
        return np.arange(base, base + self.batch_size)

    def _next(self, run):
        split = run.feeds[self.handle]
        L = int(run.feeds.get(self.chunk_size, self.default_chunk))
        b = self.cursor[split] % self.nb_batches[split]
        self.cursor[split] += 1
        key = (split, b % self.pool_batches, L)
        if key not in self.pool:
            mix, non_mix, ind = synthetic_mixtures(self._batch_indices(split, b % self.pool_batches), self.S, L)
            self.pool[key] = (torch.from_numpy(mix).to(self.device), torch.from_numpy(non_mix).to(self.device),
                              torch.from_numpy(ind).to(self.device))
        return self.pool[key]
