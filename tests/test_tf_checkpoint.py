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


# ---- a bundle index assembled BY HAND (not by tf_checkpoint.write_bundle) ------------------------------------------------------
def _crc32c_bitwise(data):
    """Castagnoli CRC, bit at a time (polynomial 0x1EDC6F41 reflected) -- independent of csrc/host/crc32c.c's slice-by-8 tables."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF          # leveldb / TF crc32c::Mask


def _trailer(block):
    return block + b'\x00' + struct.pack('<I', _mask(_crc32c_bitwise(block + b'\x00')))


def test_reader_on_a_hand_assembled_index(tmp_path):
    """The reader against bytes this repo's writer did not produce.  Layout written out literally below, the way
    tensorflow/core/util/tensor_bundle (BundleWriter) + lib/io/table_builder emit it: header value with the `version` submessage
    the real writer adds, BundleEntryProto with proto3 zero fields OMITTED (offset 0, shard 0), a scalar's EMPTY shape message,
    prefix-compressed keys ("a/bias" -> "a/kernel" shares "a/"; restart interval 16 so one restart point), a second data
    block, an index block with restart interval 1 and SHORTENED separator keys (leveldb FindShortestSeparator: "a/l" instead of
    "a/kernel"), masked CRC-32C trailers computed bit-wise here."""
    kernel = np.arange(6, dtype='<f4').reshape(1, 2, 3)                     # [1, Din, Dout] like the reference's Conv1D W
    bias = np.array([0.5, -1.0, 2.0], dtype='<f4')
    step = np.array(7, dtype='<i4')                                         # scalar
    data = bias.tobytes() + kernel.tobytes() + step.tobytes()               # offsets 0, 12, 36

    def crc_field(blob):
        return b'\x35' + struct.pack('<I', _mask(_crc32c_bitwise(blob)))     # field 6, wire type 5 (fixed32)
    header = bytes.fromhex('0801' '1a02' '0801')                             # num_shards = 1, version { producer: 1 }
    e_bias = bytes.fromhex('0801' '1204' '1202' '0803' '280c') + crc_field(bias.tobytes())            # dtype 1, shape{dim{3}}, size 12 (offset 0 omitted)
    e_kern = bytes.fromhex('0801' '120c' '1202' '0801' '1202' '0802' '1202' '0803' '200c' '2818') + crc_field(kernel.tobytes())
    e_step = bytes.fromhex('0803' '1200' '2024' '2804') + crc_field(step.tobytes())                   # dtype 3 (int32), empty shape, offset 36, size 4
    block0 = (bytes([0, 0, len(header)]) + header                           # key ""        shared 0, non_shared 0
              + bytes([0, 6, len(e_bias)]) + b'a/bias' + e_bias             # key "a/bias"  shared 0, non_shared 6
              + bytes([2, 6, len(e_kern)]) + b'kernel' + e_kern             # key "a/kernel": shares "a/"
              + struct.pack('<II', 0, 1))                                   # restart offsets [0], num_restarts 1
    block1 = (bytes([0, 11, len(e_step)]) + b'global_step' + e_step
              + struct.pack('<II', 0, 1))
    out = bytearray()
    off0 = len(out); out += _trailer(block0)
    off1 = len(out); out += _trailer(block1)
    meta = struct.pack('<II', 0, 1)                                         # empty metaindex block
    offm = len(out); out += _trailer(meta)
    assert off0 < 128 and len(block0) < 256 and off1 < 16384                # so the handles below are the 1- and 2-byte varints written
    def varint(n):
        b = bytearray()
        while n >= 0x80:
            b.append((n & 0x7f) | 0x80)
            n >>= 7
        b.append(n)
        return bytes(b)
    h0 = varint(off0) + varint(len(block0))
    h1 = varint(off1) + varint(len(block1))
    index = (bytes([0, 3, len(h0)]) + b'a/l' + h0                           # shortened separator >= "a/kernel", < "global_step"
             + bytes([0, 1, len(h1)]) + b'h' + h1                           # short successor of "global_step"
             + struct.pack('<III', 0, 3 + 3 + len(h0), 2))                  # restart interval 1: a restart per entry
    offi = len(out); out += _trailer(index)
    footer = varint(offm) + varint(len(meta)) + varint(offi) + varint(len(index))
    footer += b'\x00' * (40 - len(footer)) + bytes.fromhex('57fb808b247547db')     # kTableMagicNumber, little endian
    out += footer
    prefix = str(tmp_path / 'model-3')
    open(prefix + '.index', 'wb').write(bytes(out))
    open(prefix + '.data-00000-of-00001', 'wb').write(data)

    nshards, entries = tfc.list_bundle(prefix)
    assert nshards == 1 and sorted(entries) == ['a/bias', 'a/kernel', 'global_step']
    assert entries['a/kernel']['shape'] == [1, 2, 3] and entries['a/kernel']['offset'] == 12
    assert entries['global_step']['shape'] == [] and entries['a/bias']['offset'] == 0
    got = tfc.read_bundle(prefix)
    assert np.array_equal(got['a/kernel'], kernel) and got['a/kernel'].shape == (1, 2, 3)
    assert np.array_equal(got['a/bias'], bias) and got['global_step'] == 7 and got['global_step'].dtype == np.int32
    # and the in-repo CRC agrees with the bit-wise one on these blocks
    from data import tfrecord
    assert tfrecord.masked_crc(block0 + b'\x00') == _mask(_crc32c_bitwise(block0 + b'\x00'))


# ---- snappy-compressed table blocks (block type 1) --------------------------------------------------------------------------------
def test_snappy_decoder_known_streams():
    """The raw snappy format, element by element (format_description.txt): a hand-assembled stream with a literal, the three
    copy encodings and a copy that overlaps its own output (run-length form), then streams from an independent ENCODER (pyarrow's
    snappy codec), including a literal longer than 60 bytes (length in trailing bytes) and incompressible input."""
    stream = (bytes([23])                                  # uncompressed length 23
              + bytes([(4 - 1) << 2]) + b'abcd'            # literal "abcd"
              + bytes([0b000_001_01, 4])                   # copy-1: len 4 + 1 = 5, offset 4  -> "abcda" (overlaps: off < len)
              + bytes([((6 - 1) << 2) | 2, 9, 0])          # copy-2: len 6, offset 9         -> "abcdab"
              + bytes([((8 - 1) << 2) | 3, 2, 0, 0, 0]))   # copy-4: len 8, offset 2         -> "abababab"
    assert tfc._snappy_raw(stream) == b'abcd' + b'abcda' + b'abcdab' + b'abababab'
    pa = pytest.importorskip('pyarrow')
    rng = np.random.RandomState(0)
    for blob in (b'', b'x', b'abcabcabcabcabcabc hello hello hello' * 40, bytes(rng.randint(0, 256, 5000, dtype=np.uint8)),
                 bytes(rng.randint(0, 4, 70000, dtype=np.uint8)), b'q' * 100000):
        assert tfc._snappy_raw(pa.compress(blob, codec='snappy', asbytes=True)) == blob
    with pytest.raises(IOError):
        tfc._snappy_raw(bytes([5]) + bytes([(3 - 1) << 2]) + b'abc')          # expands to 3 bytes, header says 5
    with pytest.raises(IOError):
        tfc._snappy_raw(bytes([4]) + bytes([0b000_000_01, 9]))                 # copy before anything was produced


def test_reader_on_snappy_compressed_blocks(tmp_path):
    """A table written with block compression on: every block of a bundle index re-encoded as type 1 (snappy body, CRC over
    compressed body + type byte, as table_builder.cc does) must read back like the raw one."""
    pa = pytest.importorskip('pyarrow')
    rng = np.random.RandomState(3)
    arrays = _arrays(rng, n=40)
    prefix = str(tmp_path / 'model-5')
    tfc.write_bundle(prefix, arrays)
    raw = open(prefix + '.index', 'rb').read()
    footer = raw[-48:]
    pos = 0
    offm, pos = tfc._uvarint(footer, pos); szm, pos = tfc._uvarint(footer, pos)
    offi, pos = tfc._uvarint(footer, pos); szi, pos = tfc._uvarint(footer, pos)
    index_entries = tfc._read_block(raw, offi, szi)                          # [(separator key, handle bytes)]

    def varint(n):
        b = bytearray()
        while n >= 0x80:
            b.append((n & 0x7f) | 0x80)
            n >>= 7
        b.append(n)
        return bytes(b)

    def packed(body):
        comp = pa.compress(body, codec='snappy', asbytes=True)
        return comp + b'\x01' + struct.pack('<I', _mask(_crc32c_bitwise(comp + b'\x01'))), len(comp)
    out = bytearray()
    new_index = bytearray()
    restarts = []
    for key, handle in index_entries:
        off, size, _ = tfc._handle(handle, 0)
        blob, clen = packed(raw[off:off + size])
        h = varint(len(out)) + varint(clen)
        out += blob
        restarts.append(len(new_index))
        new_index += bytes([0]) + varint(len(key)) + varint(len(h)) + key + h
    new_index += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    meta_blob, mlen = packed(raw[offm:offm + szm])
    offm2 = len(out); out += meta_blob
    idx_blob, ilen = packed(bytes(new_index))
    offi2 = len(out); out += idx_blob
    foot = varint(offm2) + varint(mlen) + varint(offi2) + varint(ilen)
    out += foot + b'\x00' * (40 - len(foot)) + bytes.fromhex('57fb808b247547db')
    prefix2 = str(tmp_path / 'model-6')
    open(prefix2 + '.index', 'wb').write(bytes(out))
    for f in os.listdir(str(tmp_path)):
        if f.startswith('model-5.data'):
            open(os.path.join(str(tmp_path), f.replace('model-5', 'model-6')), 'wb').write(open(os.path.join(str(tmp_path), f), 'rb').read())
    got = tfc.read_bundle(prefix2)
    assert sorted(got) == sorted(arrays)
    for k in arrays:
        assert np.array_equal(got[k], arrays[k]) and got[k].dtype == arrays[k].dtype


def test_missing_variable_error_lists_what_the_bundle_holds(tmp_path):
    """The BLSTM / dense variable names of a reference checkpoint are inferred: when one is not in the bundle the
    error must name what IS there, so that a renamed variable is one edit away instead of a silent zero-fill."""
    import types
    import torch
    from models.network import Network
    folder = str(tmp_path)
    tfc.write_bundle(os.path.join(folder, 'model-3'), {'front/window/w': np.ones(4, np.float32), 'prediction/weights_typo': np.zeros((2, 2), np.float32)})
    open(os.path.join(folder, 'checkpoint'), 'w').write('model_checkpoint_path: "model-3"\n')
    fake = types.SimpleNamespace(saver=['front/window/w', 'prediction/W'])
    from ams_hip.graph import Graph
    g = Graph()
    with g.as_default():
        from ams_hip.graph import get_scope_variable  # noqa: F401
        g.variables['front/window/w'] = torch.zeros(4)
        g.variables['prediction/W'] = torch.zeros(2, 2)
        with pytest.raises(KeyError) as e:
            Network.restore_model(fake, folder)
    assert 'prediction/W' in str(e.value) and 'prediction/weights_typo' in str(e.value) and 'front/window/w' in str(e.value)
