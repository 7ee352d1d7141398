"""TFRecord files of tf.train.Example{'audio': bytes (float32 samples), 'key': int64 (speaker index)} without TensorFlow
(reference data/dataset.py:405-452 writes them with tf.python_io.TFRecordWriter, :444-452 decodes them) -- SURVEY 8f row N3.

Format (public, stable since TF 0.x):
  record   = uint64 length | uint32 masked_crc32c(length bytes) | data[length] | uint32 masked_crc32c(data)   (little endian)
  masked   = ((crc >> 15 | crc << 17) + 0xa282ead8) mod 2^32,  crc = CRC-32C (Castagnoli, reflected poly 0x82F63B78)
  data     = serialized tf.train.Example = message{ 1: Features{ 1: map<string, Feature> } },
             Feature{ 1: BytesList{1: repeated bytes} | 2: FloatList{1: packed float} | 3: Int64List{1: packed varint} }
Only the wire subset those two features need is parsed.  No TF-written file is available in this environment: the reader is
pinned by the published CRC-32C check value and by round trips through the writer below (tests/test_tfrecord.py).
"""
import struct

import numpy as np

_MASK_DELTA = 0xa282ead8


def _make_table():
    tbl = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tbl.append(c)
    return np.array(tbl, dtype=np.uint32)


_TABLE = _make_table()
_TABLE_LIST = [int(v) for v in _TABLE]


_C = None


def _native():
    """libams_host.so (csrc/host/crc32c.c, built by `make`): ~1 GB/s instead of ~10 MB/s for the pure-Python loop below."""
    global _C
    if _C is None:
        import ctypes
        import os
        path = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'ams_hip', 'libams_host.so'))
        try:
            lib = ctypes.CDLL(path)
            lib.ams_crc32c.restype = ctypes.c_uint32
            lib.ams_crc32c.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
            _C = lib
        except OSError:
            _C = False
    return _C


def crc32c(data):
    """CRC-32C of a bytes-like object (check value: crc32c(b'123456789') == 0xE3069283)."""
    lib = _native()
    if lib:
        b = bytes(data)
        return int(lib.ams_crc32c(b, len(b)))
    return crc32c_py(data)


def crc32c_py(data):
    c = 0xFFFFFFFF
    tbl = _TABLE_LIST
    for b in memoryview(data).cast('B'):
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---- protobuf wire helpers -----------------------------------------------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1                     # int64 as two's complement, like protobuf
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf, pos):
    shift = 0
    val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _ld(field, payload):                    # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    """Yield (field number, wire type, value) for one message; value is bytes for length-delimited, int for varint."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _read_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield field, wt, v


def encode_example(audio, key):
    """serialized tf.train.Example with features 'audio' (float32 bytes) and 'key' (int64) -- dataset.py:431-436."""
    audio = np.ascontiguousarray(audio, dtype=np.float32).tobytes()
    f_audio = _ld(1, _ld(1, audio))                                    # Feature.bytes_list{value}
    f_key = _ld(3, _ld(1, _varint(int(key))))                          # Feature.int64_list{packed value}
    entries = b''
    for name, feat in (('audio', f_audio), ('key', f_key)):
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))        # map entry {1: key, 2: value}
    return _ld(1, entries)                                             # Example.features


def decode_example(data):
    """-> (audio float32 array, key int) -- the reference's decode() (dataset.py:444-452)."""
    audio, key = None, None
    for f, _, features in _fields(data):
        if f != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            name, feat = None, None
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    name = v.decode()
                elif f3 == 2:
                    feat = v
            for kind, _, lst in _fields(feat or b''):
                if name == 'audio' and kind == 1:
                    for f5, _, v in _fields(lst):
                        if f5 == 1:
                            audio = np.frombuffer(v, dtype='<f4').copy()
                elif name == 'key' and kind == 3:
                    for f5, wt, v in _fields(lst):
                        if f5 == 1:
                            val = _read_varint(v, 0)[0] if wt == 2 else v
                            key = val - (1 << 64) if val >= (1 << 63) else val
    if audio is None or key is None:
        raise ValueError("example lacks the 'audio' / 'key' features")
    return audio, int(key)


# ---- record framing ---------------------------------------------------------------------------------------------------
def write_records(path, payloads):
    with open(path, 'wb') as f:
        for data in payloads:
            head = struct.pack('<Q', len(data))
            f.write(head + struct.pack('<I', masked_crc(head)) + data + struct.pack('<I', masked_crc(data)))


def read_records(path, check_crc=True):
    with open(path, 'rb') as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise IOError('truncated TFRecord header in %s' % path)
            (ln,) = struct.unpack('<Q', head)
            (hcrc,) = struct.unpack('<I', f.read(4))
            if check_crc and hcrc != masked_crc(head):
                raise IOError('corrupt TFRecord length in %s' % path)
            data = f.read(ln)
            tail = f.read(4)
            if len(data) != ln or len(tail) != 4:
                raise IOError('truncated TFRecord in %s' % path)
            if check_crc and struct.unpack('<I', tail)[0] != masked_crc(data):
                raise IOError('corrupt TFRecord payload in %s' % path)
            yield data


def write_audio_records(path, items):
    """items: iterable of (float32 audio array, speaker key)."""
    write_records(path, (encode_example(a, k) for a, k in items))


def read_audio_records(path, check_crc=True):
    for data in read_records(path, check_crc):
        yield decode_example(data)
