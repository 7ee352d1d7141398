"""Input pipeline mirror (reference data/dataset.py:532-645 ``TFDataset``).

Two sources honour the same OUTPUT contract -- ``next_mix [B,L]``, ``next_non_mix [B,S,L]``, ``next_ind [B,S]`` with
``x_mix = sum_s x_s`` exactly as dataset.py:462-468:
  * ``--dataset synthetic`` (default of the tests and benches): the generator SURVEY 8(d) specifies;
  * any other ``--dataset``: the reference's TFRecord files ``{train,valid,test,test_other}_{M,F}.tfrecords`` under
    ``config.workdir``, read without TensorFlow by data/tfrecord.py and run through the same stages
    (decode -> optional per-utterance normalise -> shuffle(100) -> keep utterances longer than the chunk -> chunk -> shuffle(10)
    -> zip S gender streams -> drop tuples with a repeated speaker -> sum -> batch), see ``RecordStream`` / ``MixtureStream``.
    tf.data's shuffle buffers draw from TF's own RNG, so the ORDER of examples differs from a TensorFlow run; the set of examples
    an epoch yields and every per-example value are the same.
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


def _shuffle_buffer(it, size, rng):
    """tf.data.Dataset.shuffle(size) semantics: keep `size` elements, emit a uniformly drawn one, refill."""
    buf = []
    for x in it:
        if len(buf) < size:
            buf.append(x)
            continue
        j = rng.randint(0, size)
        out, buf[j] = buf[j], x
        yield out
    while buf:
        j = rng.randint(0, len(buf))
        yield buf.pop(j)


class RecordStream(object):
    """One gender stream of the reference (TFDataset.get_data, dataset.py:519-527): an iterable of (chunk [L] float32, key)."""

    def __init__(self, path, chunk_size, seed, normalize=False):
        self.path, self.chunk_size, self.seed, self.normalize = path, int(chunk_size), int(seed), bool(normalize)

    def __iter__(self):
        from data import tfrecord
        rng1, rng2 = np.random.RandomState(self.seed), np.random.RandomState(self.seed + 7919)
        L = self.chunk_size

        def utterances():
            for audio, key in tfrecord.read_audio_records(self.path):
                if self.normalize:                                   # dataset.py:454-458: population moments over the utterance
                    audio = (audio - audio.mean()) / np.sqrt(audio.var())
                yield audio, key

        def chunks():
            for audio, key in _shuffle_buffer(utterances(), 100, rng1):
                if not L < audio.shape[0]:                           # is_long_enough: chunk_size < length (strict, :470-471)
                    continue
                for i in range(audio.shape[0] // L):                 # chunk(): floor(L_utt / chunk) pieces (:482-491)
                    yield audio[i * L:(i + 1) * L], key
        return _shuffle_buffer(chunks(), 10, rng2)


class MixtureStream(object):
    """zip of S RecordStreams -> filter distinct speakers (dataset.py:473-480) -> mix (:462-468): an iterable of
    (mix [L], non_mix [S, L], keys [S]) examples."""

    def __init__(self, streams):
        self.streams = streams

    def __iter__(self):
        for items in zip(*self.streams):
            ks = [k for _, k in items]
            if len(set(ks)) != len(ks):
                continue
            nm = np.stack([c for c, _ in items]).astype(np.float32)
            yield nm.sum(axis=0), nm, np.asarray(ks, np.int32)


def round_robin(streams):
    """`process()` of the reference (dataset.py:493-511), its DEFAULT branch: zip the per-combination mixture streams, stack,
    unbatch -- i.e. one example of combination 0, one of combination 1, ... and over again, ending with the first exhausted
    combination (tf.data zip semantics; a partially filled round is dropped because zip yields whole tuples only)."""
    for round_ in zip(*streams):
        for ex in round_:
            yield ex


def batched(examples, batch_size, drop_remainder=False):
    mixes, nms, keys = [], [], []
    for m, nm, k in examples:
        mixes.append(m)
        nms.append(nm)
        keys.append(k)
        if len(mixes) == batch_size:
            yield np.stack(mixes), np.stack(nms), np.stack(keys)
            mixes, nms, keys = [], [], []
    if mixes and not drop_remainder:                                  # tf.data batch() keeps the short final batch
        yield np.stack(mixes), np.stack(nms), np.stack(keys)


def record_mixture_stream(folder, split, sex, S, chunk_size, batch_size, normalize=False, no_random_picking=True, epoch=0,
                          drop_remainder=False):
    """TFDataset.__init__ of the reference (dataset.py:544-628), both branches:
      * `no_random_picking` (:561-566,586-605): speaker i comes from the M file when i is even and the F file when odd (both
        genders requested), stream i seeded with i; single-gender runs (:576-585) take every speaker from that file;
      * default (:567-575,625-628): one mixture stream per gender combination product([M, F], repeat=S) -- [M,M] [M,F] [F,M]
        [F,F] for two speakers -- stream j of combination i seeded with j + S*i, interleaved example by example (`process`).
    tf.data's shuffle(seed) reshuffles on every initialisation of the iterator; `epoch` (how many times the split has been
    initialised) is folded into the shuffle seeds for the same effect."""
    import os
    from itertools import product
    bump = 104729 * int(epoch)

    def stream(g, seed):
        return RecordStream(os.path.join(folder, '%s_%s.tfrecords' % (split, g)), chunk_size, seed + bump, normalize)
    both = 'M' in sex and 'F' in sex
    if both and not no_random_picking:
        combos = [MixtureStream([stream(g, j + S * i) for j, g in enumerate(comb)])
                  for i, comb in enumerate(product(['M', 'F'], repeat=S))]
        examples = round_robin(combos)
    elif both:
        examples = MixtureStream([stream('M' if i % 2 == 0 else 'F', i) for i in range(S)])
    else:
        g = 'M' if 'M' in sex else 'F'
        examples = MixtureStream([stream(g, i) for i in range(S)])
    return batched(examples, batch_size, drop_remainder)


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
        # real data: the reference's TFRecord files next to config.workdir (or AMS_DATA_DIR)
        self.records = None
        name = kwargs.get('dataset')
        if name and name != 'synthetic':
            import os
            import config
            folder = os.environ.get('AMS_DATA_DIR', config.workdir)
            if not os.path.exists(os.path.join(folder, 'train_M.tfrecords')) and not os.path.exists(os.path.join(folder, 'train_F.tfrecords')):
                raise IOError('--dataset %s: no {split}_{M,F}.tfrecords under %s (set AMS_DATA_DIR, or use --dataset synthetic)'
                              % (name, folder))
            self.records = dict(folder=folder, sex=kwargs.get('sex') or ['M', 'F'], normalize=bool(kwargs.get('dataset_normalize')),
                                no_random_picking=bool(kwargs.get('no_random_picking', False)))
            self._iters = {}
            self._lengths = {}
            self._epochs = {}
            self.hip_graph = bool(kwargs.get('hip_graph'))

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
        if self.records is not None:
            self._iters.pop(split, None)
            self._epochs[split] = self._epochs.get(split, -1) + 1       # reshuffle_each_iteration (tf.data default)

    def _drop_remainder(self, split):
        """A short final batch cannot be replayed through a hipGraph captured at the full batch shape, and with N ranks it would
        reach one rank only while the 1/world gradient scale assumes equal shards: dropped in those cases (kept otherwise, as
        tf.data's batch() does)."""
        world = self.dist.world_size if self.dist is not None else 1
        return world > 1 or (self.hip_graph and split == self.TRAIN)

    def _stream(self, split, L, epoch):
        r = self.records
        return record_mixture_stream(r['folder'], split, r['sex'], self.S, L, self.batch_size, r['normalize'],
                                     r['no_random_picking'], epoch, self._drop_remainder(split))

    def _record_batch(self, split, L):
        """Next batch of the TFRecord pipeline; with N ranks each rank keeps every N-th batch (utterance-level sharding)."""
        world = self.dist.world_size if self.dist is not None else 1
        rank = self.dist.rank if self.dist is not None else 0
        out = None
        for k in range(world):
            it = self._iters.get(split)
            if it is None:
                it = self._iters[split] = iter(self._stream(split, L, self._epochs.get(split, 0)))
            try:
                b = next(it)
            except StopIteration:
                # End of the pass = tf.errors.OutOfRangeError.  One exception: length() counted a pass in epoch-0 order and a
                # RESHUFFLED pass can come out a batch short (the distinct-speaker filter depends on the order); the reference
                # would die mid-epoch there.  While the caller is still inside the counted epoch, continue with the next pass.
                if self.cursor[split] >= self._lengths.get((split, L), 0):
                    raise
                self._epochs[split] = self._epochs.get(split, 0) + 1
                it = self._iters[split] = iter(self._stream(split, L, self._epochs[split]))
                b = next(it)
            if k == rank:
                out = b
        self.cursor[split] += 1
        return tuple(torch.from_numpy(a).to(self.device) for a in out)

    def length(self, split):
        if self.records is not None:
            # dataset.py:667-676: count the batches of one pass over the split (cached); each rank sees every world-th batch
            key = (split, self.default_chunk)
            if key not in self._lengths:
                n = sum(1 for _ in self._stream(split, self.default_chunk, 0))
                world = self.dist.world_size if self.dist is not None else 1
                self._lengths[key] = n // world
            return self._lengths[key]
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
        if self.records is not None:
            return self._record_batch(split, L)
        b = self.cursor[split] % self.nb_batches[split]
        self.cursor[split] += 1
        key = (split, b % self.pool_batches, L)
        if key not in self.pool:
            mix, non_mix, ind = synthetic_mixtures(self._batch_indices(split, b % self.pool_batches), self.S, L)
            # mixtures and sources back to back in ONE device buffer: a captured step then stages a batch with one copy instead of two
            # (models/network.py::_train_graphed)
            flat = torch.from_numpy(np.concatenate([mix.ravel(), non_mix.ravel()])).to(self.device)
            self.pool[key] = (flat[:mix.size].view(mix.shape), flat[mix.size:].view(non_mix.shape), torch.from_numpy(ind).to(self.device))
        return self.pool[key]
