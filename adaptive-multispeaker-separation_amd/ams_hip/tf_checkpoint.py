"""TensorFlow "V2" checkpoint bundles (``<prefix>.index`` + ``<prefix>.data-00000-of-0000N``) without TensorFlow -- SURVEY 8f row N2.

Lets a pretrained reference model (saved by tf.train.Saver, reference models/network.py:125-145,223-226) be restored into this
build, and lets this build export weights the reference can load.  UNVERIFIED against a TensorFlow-written file: none exists in
this environment, so the reader follows the published formats below and is tested by round trips through the writer in this
module and by structural checks (tests/test_tf_checkpoint.py).  Variable names are the graph's own (``front/window/w``,
``prediction/forward_BLSTM_0/rnn/basic_lstm_cell/kernel`` ...): the host mirror reproduces the reference's scopes
(utils/ops.py:371-380, models/network.py:524,611).

``.index`` is a LevelDB-format table (tensorflow/core/lib/io/table*, same layout as leveldb's table_format.md):
  [data blocks][metaindex block][index block][footer 48 B = metaindex handle, index handle (varint64 offset,size), pad, magic]
  block  = entries (shared:varint32, non_shared:varint32, value_len:varint32, key delta, value) + restart offsets (uint32 each)
           + num_restarts (uint32); followed on disk by 1 type byte (0 = raw, 1 = snappy) + 4 B masked CRC-32C of block+type.
  keys   = "" -> BundleHeaderProto{1 num_shards, 2 endianness, 3 version}; variable name -> BundleEntryProto{1 dtype, 2 shape
           (TensorShapeProto{2 dim{1 size}}), 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32, masked)}.
``.data-*`` holds the raw little-endian tensor bytes at [offset, offset+size).
BundleWriter writes raw blocks; a table written with snappy block compression (type 1) is decoded by `_snappy_raw` below
(the raw snappy format: varint32 uncompressed length, then literal / copy elements), so such an index loads as well.
"""
import os
import struct

import numpy as np

from data import tfrecord as _tr

_MAGIC = 0xdb4775248b80fb57
_DT = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 4: np.uint8, 6: np.int8, 5: np.int16, 10: np.bool_}
_DT_INV = {np.dtype(v): k for k, v in _DT.items()}


def _snappy_raw(buf):
    """Decoder of the raw snappy format (format_description.txt of the snappy library): preamble = uncompressed length as a
    varint32; elements tagged by the low two bits of their first byte: 00 literal (length-1 in the upper six bits, or in the next
    1-4 bytes for 60-63), 01 copy with 11-bit offset (length 4-11), 10 copy with 16-bit offset, 11 copy with 32-bit offset
    (length 1-64).  Copies may overlap their own output (run-length form), so they are expanded byte-wise when they do."""
    n, pos = _uvarint(buf, 0)
    out = bytearray()
    L = len(buf)
    while pos < L:
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            if pos + ln > L:
                raise IOError('snappy: literal runs past the end of the block')
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise IOError('snappy: copy offset %d outside the %d bytes produced so far' % (off, len(out)))
        start = len(out) - off
        if off >= ln:
            out += out[start:start + ln]
        else:
            for i in range(ln):
                out.append(out[start + i])
    if len(out) != n:
        raise IOError('snappy: block expands to %d bytes, header says %d' % (len(out), n))
    return bytes(out)


def _uvarint(buf, pos):
    return _tr._read_varint(buf, pos)


def _handle(buf, pos):
    off, pos = _uvarint(buf, pos)
    size, pos = _uvarint(buf, pos)
    return off, size, pos


def _read_block(raw, off, size, verify=True):
    body = raw[off:off + size]
    typ = raw[off + size]
    (crc,) = struct.unpack('<I', raw[off + size + 1:off + size + 5])
    if verify and crc != _tr.masked_crc(raw[off:off + size + 1]):
        raise IOError('TF checkpoint index: block CRC mismatch at offset %d' % off)
    if typ == 1:
        body = _snappy_raw(body)
    elif typ != 0:
        raise NotImplementedError('TF checkpoint index: block compression type %d (0 = raw and 1 = snappy are defined)' % typ)
    (nrestart,) = struct.unpack('<I', body[-4:])
    end = len(body) - 4 * (nrestart + 1)
    pos, key = 0, b''
    out = []
    while pos < end:
        shared, pos = _uvarint(body, pos)
        nonshared, pos = _uvarint(body, pos)
        vlen, pos = _uvarint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + nonshared])
        pos += nonshared
        out.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
    return out


def _parse_entry(value):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None}
    for f, wt, v in _tr._fields(value):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for f2, _, dim in _tr._fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, dv in _tr._fields(dim):
                        if f3 == 1:
                            size = dv
                    e['shape'].append(int(size))
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = struct.unpack('<I', v)[0]
    return e


def list_bundle(prefix, verify=True):
    """-> (num_shards, {name: entry dict}) from ``<prefix>.index``."""
    raw = open(prefix + '.index', 'rb').read()
    if len(raw) < 48 or struct.unpack('<Q', raw[-8:])[0] != _MAGIC:
        raise IOError('%s.index is not a TensorFlow V2 checkpoint index (bad magic)' % prefix)
    footer = raw[-48:]
    _, _, pos = _handle(footer, 0)                       # metaindex (unused)
    ioff, isize, _ = _handle(footer, pos)
    entries, num_shards = {}, 1
    for _, hv in _read_block(raw, ioff, isize, verify):
        doff, dsize, _ = _handle(hv, 0)
        for key, value in _read_block(raw, doff, dsize, verify):
            if key == b'':
                for f, _, v in _tr._fields(value):
                    if f == 1:
                        num_shards = v
            else:
                entries[key.decode()] = _parse_entry(value)
    return num_shards, entries


def read_bundle(prefix, names=None, verify=True):
    """{variable name: numpy array} for all (or the requested) variables of a V2 checkpoint."""
    num_shards, entries = list_bundle(prefix, verify)
    shards = {}
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['dtype'] not in _DT:
            continue                                      # strings / resources: not model weights
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), dtype=np.uint8, mode='r')
        blob = bytes(shards[sid][e['offset']:e['offset'] + e['size']])
        if verify and e['crc32c'] is not None and e['crc32c'] not in (_tr.masked_crc(blob), _tr.crc32c(blob)):
            raise IOError('TF checkpoint: tensor %s fails its CRC' % name)
        out[name] = np.frombuffer(blob, dtype=np.dtype(_DT[e['dtype']]).newbyteorder('<')).reshape(e['shape']).copy()
    return out


# ---- writer (export; also what the tests read back) ----------------------------------------------------------------------
def _block(entries, restart_interval=16):
    body = bytearray()
    restarts = []
    last = b''
    for i, (key, value) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            n = min(len(last), len(key))
            while shared < n and last[shared] == key[shared]:
                shared += 1
        body += _tr._varint(shared) + _tr._varint(len(key) - shared) + _tr._varint(len(value)) + key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack('<I', r)
    body += struct.pack('<I', len(restarts))
    return bytes(body)


def _emit(out, body):
    off = len(out)
    out += body + b'\x00' + struct.pack('<I', _tr.masked_crc(body + b'\x00'))
    return off, len(body)


def write_bundle(prefix, arrays):
    """Write {name: array} as a single-shard V2 checkpoint (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(arrays)
    data = bytearray()
    entries = [(b'', _tr._varint((1 << 3) | 0) + _tr._varint(1))]                        # BundleHeaderProto{num_shards = 1}
    for n in names:
        a = np.asarray(arrays[n])                        # (ascontiguousarray would turn a scalar into shape (1,))
        blob = a.astype(a.dtype.newbyteorder('<')).tobytes()
        shape = b''.join(_tr._ld(2, _tr._varint((1 << 3) | 0) + _tr._varint(int(d))) for d in a.shape)
        e = (_tr._varint((1 << 3) | 0) + _tr._varint(_DT_INV[np.dtype(a.dtype)]) + _tr._ld(2, shape)
             + _tr._varint((4 << 3) | 0) + _tr._varint(len(data)) + _tr._varint((5 << 3) | 0) + _tr._varint(len(blob))
             + _tr._varint((6 << 3) | 5) + struct.pack('<I', _tr.masked_crc(blob)))
        entries.append((n.encode(), e))
        data += blob
    out = bytearray()
    index_entries = []
    per = 64
    for i in range(0, len(entries), per):
        chunk = entries[i:i + per]
        off, size = _emit(out, _block(chunk))
        index_entries.append((chunk[-1][0], _tr._varint(off) + _tr._varint(size)))
    moff, msize = _emit(out, _block([]))
    ioff, isize = _emit(out, _block(index_entries, restart_interval=1))
    footer = _tr._varint(moff) + _tr._varint(msize) + _tr._varint(ioff) + _tr._varint(isize)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC)
    out += footer
    open(prefix + '.index', 'wb').write(bytes(out))
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))


def reference_shapes(arrays):
    """{name: array} as the REFERENCE's graph declares them: its Conv1D kernels (`prediction/W`, `enhance/W`) are 3-D
    [1, Din, Dout] (utils/ops.py:486-492) where this build keeps [Din, Dout].  Use before write_bundle when exporting a model the
    reference should load; Network.restore_model squeezes the leading 1 on the way in."""
    out = {}
    for n, a in arrays.items():
        a = np.asarray(a)
        out[n] = a[None] if (n.endswith('/W') and a.ndim == 2) else a
    return out


def latest_checkpoint(folder):
    """tf.train.latest_checkpoint: parse the text ``checkpoint`` file (model_checkpoint_path: "model-123")."""
    marker = os.path.join(folder, 'checkpoint')
    if not os.path.exists(marker):
        return None
    for line in open(marker):
        line = line.strip()
        if line.startswith('model_checkpoint_path:'):
            name = line.split(':', 1)[1].strip().strip('"')
            return name if os.path.isabs(name) else os.path.join(folder, name)
    return None
