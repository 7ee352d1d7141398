"""CPU: TFRecord framing / tf.train.Example subset without TensorFlow and the reference's pipeline stages."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd'))

from data import tfrecord  # noqa: E402


def test_crc32c_check_values():
    # published CRC-32C (Castagnoli) check value and the RFC 3720 all-zero / all-one 32-byte vectors
    assert tfrecord.crc32c_py(b'123456789') == 0xE3069283
    assert tfrecord.crc32c_py(bytes(32)) == 0x8A9136AA
    assert tfrecord.crc32c_py(b'\xff' * 32) == 0x62A8AB43
    data = np.random.RandomState(0).bytes(10007)
    assert tfrecord.crc32c(data) == tfrecord.crc32c_py(data)            # native helper (if built) agrees with the pure-Python loop


def test_example_round_trip_and_corruption(tmp_path):
    rng = np.random.RandomState(1)
    items = [(rng.randn(n).astype(np.float32), k) for n, k in ((5000, 0), (1, 250), (33333, 7), (0, 3))]
    p = str(tmp_path / 'x.tfrecords')
    tfrecord.write_audio_records(p, items)
    out = list(tfrecord.read_audio_records(p))
    assert len(out) == len(items)
    for (a, k), (b, kk) in zip(items, out):
        assert np.array_equal(a, b) and k == kk
    # hand-assembled wire bytes of a minimal Example decode the same way (independent of encode_example)
    raw = np.array([1.5, -2.0], np.float32).tobytes()
    feat_a = b'\x0a' + bytes([len(raw) + 2]) + b'\x0a' + bytes([len(raw)]) + raw            # Feature{bytes_list{value}}
    feat_k = b'\x1a\x03' + b'\x0a\x01\x05'                                                # Feature{int64_list{packed [5]}}
    ent_a = b'\x0a\x05audio' + b'\x12' + bytes([len(feat_a)]) + feat_a
    ent_k = b'\x0a\x03key' + b'\x12' + bytes([len(feat_k)]) + feat_k
    feats = b'\x0a' + bytes([len(ent_a)]) + ent_a + b'\x0a' + bytes([len(ent_k)]) + ent_k
    ex = b'\x0a' + bytes([len(feats)]) + feats
    a, k = tfrecord.decode_example(ex)
    assert np.array_equal(a, [1.5, -2.0]) and k == 5
    # a flipped payload byte is caught by the masked CRC
    blob = bytearray(open(p, 'rb').read())
    blob[40] ^= 0x01
    open(p, 'wb').write(bytes(blob))
    with pytest.raises(IOError):
        list(tfrecord.read_audio_records(p))


def test_pipeline_stages_follow_the_reference(tmp_path):
    """decode -> keep utterances longer than the chunk -> floor(len/chunk) chunks -> zip speakers -> distinct speakers -> sum -> batch
    (dataset.py:462-491,519-527,586-605)."""
    from data.dataset import record_mixture_stream
    rng = np.random.RandomState(2)
    L = 100
    lens_m, lens_f = [250, 100, 99, 420], [301, 180, 1000]
    M = [(rng.randn(n).astype(np.float32), 10 + i) for i, n in enumerate(lens_m)]
    Fm = [(rng.randn(n).astype(np.float32), 20 + i) for i, n in enumerate(lens_f)]
    tfrecord.write_audio_records(str(tmp_path / 'train_M.tfrecords'), M)
    tfrecord.write_audio_records(str(tmp_path / 'train_F.tfrecords'), Fm)
    batches = list(record_mixture_stream(str(tmp_path), 'train', ['M', 'F'], 2, L, 4))
    n_m = sum(n // L for n in lens_m if L < n)               # 2 + 0 (len 100 is not > 100) + 0 + 4
    n_f = sum(n // L for n in lens_f if L < n)
    assert n_m == 6 and n_f == 14
    total = sum(b[0].shape[0] for b in batches)
    assert total == min(n_m, n_f)                             # zip stops at the shorter stream; keys differ so nothing is filtered
    assert [b[0].shape[0] for b in batches] == [4, 2]         # short final batch kept
    chunks_m = {a[i * L:(i + 1) * L].tobytes() for a, _ in M if L < a.shape[0] for i in range(a.shape[0] // L)}
    for mix, nm, keys in batches:
        assert nm.shape[1:] == (2, L) and keys.dtype == np.int32
        assert np.allclose(mix, nm.sum(axis=1))
        assert all(10 <= k < 20 for k in keys[:, 0]) and all(20 <= k < 30 for k in keys[:, 1])
        assert all(row.tobytes() in chunks_m for row in nm[:, 0])
    # same-gender streams with one speaker only: every tuple repeats the key and is dropped
    tfrecord.write_audio_records(str(tmp_path / 'valid_M.tfrecords'), [(rng.randn(500).astype(np.float32), 1)] * 3)
    assert list(record_mixture_stream(str(tmp_path), 'valid', ['M'], 2, L, 4)) == []


def test_default_branch_interleaves_gender_combinations(tmp_path):
    """README.md:23's `--men --women` command takes the DEFAULT branch (dataset.py:567-575,493-511,625-628): one mixture stream per
    gender combination [M,M] [M,F] [F,M] [F,F], stream j of combination i seeded j + S*i, examples taken round-robin over the
    combinations until the first one runs out, then batched."""
    from data.dataset import record_mixture_stream
    rng = np.random.RandomState(3)
    L, S = 50, 2
    M = [(rng.randn(50 * n + 7).astype(np.float32), 100 + i) for i, n in enumerate([3, 2, 4, 2, 3])]       # 14 chunks, 5 speakers
    Fm = [(rng.randn(50 * n + 3).astype(np.float32), 200 + i) for i, n in enumerate([2, 5, 3, 2])]         # 12 chunks, 4 speakers
    tfrecord.write_audio_records(str(tmp_path / 'train_M.tfrecords'), M)
    tfrecord.write_audio_records(str(tmp_path / 'train_F.tfrecords'), Fm)
    batches = list(record_mixture_stream(str(tmp_path), 'train', ['M', 'F'], S, L, 4, no_random_picking=False))
    keys = np.concatenate([b[2] for b in batches])
    gender = (keys >= 200).astype(int)                                  # 0 = M, 1 = F
    expect = np.array([[0, 0], [0, 1], [1, 0], [1, 1]])                 # product([M, F], repeat=2) order
    assert len(keys) % 4 == 0 and len(keys) >= 8                         # whole rounds only
    assert np.array_equal(gender, np.tile(expect, (len(keys) // 4, 1)))
    assert all(k[0] != k[1] for k in keys)                               # distinct speakers inside a mixture
    for mix, nm, _ in batches:
        assert np.allclose(mix, nm.sum(axis=1))
    # same seeds -> same pass; another epoch -> another order of the same material
    again = list(record_mixture_stream(str(tmp_path), 'train', ['M', 'F'], S, L, 4, no_random_picking=False))
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(batches, again))
    other = list(record_mixture_stream(str(tmp_path), 'train', ['M', 'F'], S, L, 4, no_random_picking=False, epoch=1))
    assert not all(np.array_equal(a[1], b[1]) for a, b in zip(batches, other))
    # drop_remainder (hipGraph replay / N ranks): only full batches
    full = list(record_mixture_stream(str(tmp_path), 'train', ['M', 'F'], S, L, 3, no_random_picking=False, drop_remainder=True))
    assert full and all(b[0].shape[0] == 3 for b in full)
