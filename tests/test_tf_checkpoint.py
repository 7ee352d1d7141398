"""CPU: TensorFlow V2 checkpoint bundle reader / writer without TensorFlow.  No TF-written file exists in this
environment: these are round trips plus structural checks against the published table format."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'adaptive-multispeaker-separation_amd'))

from ams_hip import tf_checkpoint as tfc  # noqa: E402


def _arrays(rng, n=150):
    a = {'front/window/w': rng.randn(1024).astype(np.float32), 'front/bases/bases': rng.randn(64, 16).astype(np.float32),
         'global_epoch': np.array(3, np.int32), 'prediction/b': np.zeros(0, np.float32)}
    for i in range(n):                                      # enough keys for several data blocks and shared key prefixes
        a['prediction/forward_BLSTM_%d/rnn/basic_lstm_cell/kernel' % i] = rng.randn(5, 3).astype(np.float32)
        a['prediction/forward_BLSTM_%d/rnn/basic_lstm_cell/bias' % i] = rng.randn(7).astype(np.float64)
    return a


def test_bundle_round_trip_and_structure(tmp_path):
    rng = np.random.RandomState(0)
    arrays = _arrays(rng)
    prefix = str(tmp_path / 'model-12')
    tfc.write_bundle(prefix, arrays)
    raw = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57            # leveldb table magic
    assert os.path.getsize(prefix + '.data-00000-of-00001') == sum(a.nbytes for a in arrays.values())
    nshards, entries = tfc.list_bundle(prefix)
    assert nshards == 1 and set(entries) == set(arrays)
    assert entries['front/bases/bases']['shape'] == [64, 16] and entries['front/bases/bases']['dtype'] == 1
    back = tfc.read_bundle(prefix)
    for k, v in arrays.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    sub = tfc.read_bundle(prefix, names={'front/window/w'})
    assert list(sub) == ['front/window/w']


def test_corruption_is_detected(tmp_path):
    rng = np.random.RandomState(1)
    prefix = str(tmp_path / 'model-1')
    tfc.write_bundle(prefix, _arrays(rng, 5))
    blob = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    blob[10] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(blob))
    with pytest.raises(IOError):
        tfc.read_bundle(prefix)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[5] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(IOError):
        tfc.list_bundle(prefix)
    open(prefix + '.index', 'wb').write(b'not a table')
    with pytest.raises(IOError):
        tfc.list_bundle(prefix)


def test_latest_checkpoint_text_file(tmp_path):
    (tmp_path / 'checkpoint').write_text('model_checkpoint_path: "model-900"\nall_model_checkpoint_paths: "model-100"\n')
    assert tfc.latest_checkpoint(str(tmp_path)) == os.path.join(str(tmp_path), 'model-900')
