"""TensorBoard event files without TensorFlow — what `tf.summary.FileWriter(cfg.LOGS_DIR)` + `tf.summary.merge([...])` leave on
disk for the reference's trainers (reference models/wgancls/trainer.py:20-47,104-107; models/gancls/trainer.py define_summaries).

The file is a sequence of TFRecord frames, each holding one `Event` protocol buffer:

    frame  = uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)      (little endian)
    Event  = {1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary summary}
    Summary.Value = {1: string tag, 2: float simple_value | 4: Image image | 5: HistogramProto histo}
    Image  = {1: height, 2: width, 3: colorspace, 4: bytes encoded_image_string (PNG)}
    HistogramProto = {1: min, 2: max, 3: num, 4: sum, 5: sum_squares, 6: packed double bucket_limit, 7: packed double bucket}

Semantics kept from TF 1.x: `scalar` tags are the names; `image` writes at most `max_outputs` (3) images of the batch as
`<name>/image/<i>`, float images normalised PER IMAGE (all values >= 0: largest -> 255; otherwise 0.0 -> 128 and the largest
magnitude -> +-127); `histogram` uses TF's default bucket limits (+-1e-12 * 1.1^k up to 1e20, 0 and +-DBL_MAX) with runs of empty
buckets collapsed.  Everything here is host code on NumPy arrays (tensors are brought over by the caller): it runs every
SUMMARY_PERIOD iterations, outside the hot path."""
import os
import socket
import struct
import time
import zlib

import numpy as np

# ---- CRC-32C (Castagnoli), table driven -----------------------------------------------------------------------------------
_CRC_TABLE = []


def _crc_table():
    if not _CRC_TABLE:
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    return _CRC_TABLE


def crc32c(data):
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protocol buffer wire format (the five field kinds these messages use) --------------------------------------------------
def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _f_varint(field, v):
    return _key(field, 0) + _varint(int(v))


def _f_double(field, v):
    return _key(field, 1) + struct.pack('<d', float(v))


def _f_float(field, v):
    return _key(field, 5) + struct.pack('<f', float(v))


def _f_bytes(field, b):
    b = b.encode('utf-8') if isinstance(b, str) else bytes(b)
    return _key(field, 2) + _varint(len(b)) + b


def _f_packed_doubles(field, values):
    return _f_bytes(field, struct.pack('<%dd' % len(values), *values)) if len(values) else b''


# ---- PNG (8-bit grey / RGB / RGBA, no interlace) ------------------------------------------------------------------------------
def encode_png(img):
    """img: uint8 [H, W, C] with C in (1, 3, 4)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    colour = {1: 0, 3: 2, 4: 6}[c]

    def chunk(kind, payload):
        body = kind + payload
        return struct.pack('>I', len(payload)) + body + struct.pack('>I', zlib.crc32(body) & 0xFFFFFFFF)
    rows = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * c)], axis=1)      # filter type 0 in front of every row
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, colour, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(rows.tobytes(), 6)) + chunk(b'IEND', b''))


# ---- the three summary kinds ----------------------------------------------------------------------------------------------
def scalar(tag, value):
    return _f_bytes(1, _f_bytes(1, tag) + _f_float(2, value))


def normalize_image(img):
    """One float image -> uint8, tf.summary.image's rule (see the module docstring); integer images pass through."""
    img = np.asarray(img)
    if img.dtype == np.uint8:
        return img
    x = img.astype(np.float32)
    finite = np.isfinite(x)
    if not finite.any():
        return np.zeros(x.shape, np.uint8)
    lo, hi = float(x[finite].min()), float(x[finite].max())
    tiny = 1e-6
    if lo < 0:
        m = max(abs(lo), abs(hi))
        scale, offset = (0.0 if m < tiny else 127.0 / m), 128.0
    else:
        scale, offset = (0.0 if hi < tiny else 255.0 / hi), 0.0
    y = np.where(finite, x * scale + offset, 0.0)
    return np.clip(y, 0.0, 255.0).astype(np.uint8)


def image(tag, batch, max_outputs=3):
    """batch: [N, H, W, C] (NHWC as model.x / model.G), C in (1, 3, 4)."""
    batch = np.asarray(batch)
    assert batch.ndim == 4 and batch.shape[3] in (1, 3, 4), batch.shape
    n = min(int(batch.shape[0]), max_outputs)
    out = b''
    for i in range(n):
        png = encode_png(normalize_image(batch[i]))
        h, w, c = batch.shape[1:]
        img = _f_varint(1, h) + _f_varint(2, w) + _f_varint(3, c) + _f_bytes(4, png)
        name = '%s/image/%d' % (tag, i) if max_outputs > 1 else '%s/image' % tag
        out += _f_bytes(1, _f_bytes(1, name) + _f_bytes(4, img))
    return out


_LIMITS = []


def default_bucket_limits():
    if not _LIMITS:
        pos = []
        v = 1e-12
        while v < 1e20:
            pos.append(v)
            v *= 1.1
        pos.append(float(np.finfo(np.float64).max))
        _LIMITS.extend([-p for p in reversed(pos)] + [0.0] + pos)
    return _LIMITS


def histogram(tag, values):
    v = np.asarray(values, dtype=np.float64).reshape(-1)
    limits = np.asarray(default_bucket_limits())
    # a value goes into the first bucket whose limit is greater than it
    idx = np.minimum(np.searchsorted(limits, v, side='right'), len(limits) - 1)
    counts = np.bincount(idx, minlength=len(limits)).astype(np.float64)
    lim_out, cnt_out = [], []
    i, n = 0, len(limits)
    while i < n:
        end, count = limits[i], counts[i]
        i += 1
        if count <= 0.0:
            while i < n and counts[i] <= 0.0:       # a run of empty buckets becomes one
                end, count = limits[i], counts[i]
                i += 1
        lim_out.append(float(end))
        cnt_out.append(float(count))
    h = (_f_double(1, v.min() if v.size else 0.0) + _f_double(2, v.max() if v.size else 0.0) + _f_double(3, v.size) +
         _f_double(4, v.sum()) + _f_double(5, (v * v).sum()) + _f_packed_doubles(6, lim_out) + _f_packed_doubles(7, cnt_out))
    return _f_bytes(1, _f_bytes(1, tag) + _f_bytes(5, h))


class FileWriter(object):
    """tf.summary.FileWriter(logdir): `events.out.tfevents.<seconds>.<host>` opened at construction with the version record;
    add_summary(values, step) appends one Event holding the concatenated Summary.Value records built by scalar / image /
    histogram above (the counterpart of `writer.add_summary(sess.run(summary_op), idx)`)."""

    def __init__(self, logdir, filename_suffix=''):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s%s' % (int(time.time()), socket.gethostname(), filename_suffix))
        self._f = open(self.path, 'ab')
        self._record(_f_double(1, time.time()) + _f_bytes(3, 'brain.Event:2'))
        self.flush()

    def _record(self, data):
        head = struct.pack('<Q', len(data))
        self._f.write(head + struct.pack('<I', masked_crc32c(head)) + data + struct.pack('<I', masked_crc32c(data)))

    def add_summary(self, values, global_step):
        summary = values if isinstance(values, (bytes, bytearray)) else b''.join(values)
        self._record(_f_double(1, time.time()) + _f_varint(2, int(global_step)) + _f_bytes(5, summary))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


# ---- reader (tests, and anyone who wants the scalars back without TensorBoard) --------------------------------------------------
def _parse(buf):
    """Generic wire-format walk: [(field, wire, value)], value = int / 8 raw bytes / bytes / 4 raw bytes."""
    out, i = [], 0
    while i < len(buf):
        k, sh = 0, 0
        while True:
            b = buf[i]; i += 1
            k |= (b & 0x7F) << sh; sh += 7
            if not b & 0x80:
                break
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, sh = 0, 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
        elif wire == 1:
            v = buf[i:i + 8]; i += 8
        elif wire == 5:
            v = buf[i:i + 4]; i += 4
        elif wire == 2:
            n, sh = 0, 0
            while True:
                b = buf[i]; i += 1
                n |= (b & 0x7F) << sh; sh += 7
                if not b & 0x80:
                    break
            v = buf[i:i + n]; i += n
        else:
            raise ValueError('wire type %d' % wire)
        out.append((field, wire, v))
    return out


def read_events(path):
    """-> [{'wall_time', 'step', 'file_version'?, 'values': [{'tag', 'simple_value'? | 'image'? | 'histo'?}]}]; checks both CRCs
    of every frame."""
    events = []
    with open(path, 'rb') as f:
        raw = f.read()
    i = 0
    while i < len(raw):
        head = raw[i:i + 8]
        n, = struct.unpack('<Q', head)
        if struct.unpack('<I', raw[i + 8:i + 12])[0] != masked_crc32c(head):
            raise ValueError('length CRC mismatch at %d' % i)
        data = raw[i + 12:i + 12 + n]
        if struct.unpack('<I', raw[i + 12 + n:i + 16 + n])[0] != masked_crc32c(data):
            raise ValueError('data CRC mismatch at %d' % i)
        i += 16 + n
        ev = {'values': []}
        for field, wire, v in _parse(data):
            if field == 1:
                ev['wall_time'] = struct.unpack('<d', v)[0]
            elif field == 2:
                ev['step'] = v
            elif field == 3:
                ev['file_version'] = v.decode()
            elif field == 5:
                for f2, _, val in _parse(v):
                    if f2 != 1:
                        continue
                    rec = {}
                    for f3, _, x in _parse(val):
                        if f3 == 1:
                            rec['tag'] = x.decode()
                        elif f3 == 2:
                            rec['simple_value'] = struct.unpack('<f', x)[0]
                        elif f3 == 4:
                            im = {}
                            for f4, _, y in _parse(x):
                                im[{1: 'height', 2: 'width', 3: 'colorspace', 4: 'png'}[f4]] = y
                            rec['image'] = im
                        elif f3 == 5:
                            h = {}
                            for f4, _, y in _parse(x):
                                name = {1: 'min', 2: 'max', 3: 'num', 4: 'sum', 5: 'sum_squares', 6: 'bucket_limit', 7: 'bucket'}[f4]
                                h[name] = list(struct.unpack('<%dd' % (len(y) // 8), y)) if f4 >= 6 else struct.unpack('<d', y)[0]
                            rec['histo'] = h
                    ev['values'].append(rec)
        events.append(ev)
    return events


def decode_png(png):
    """Inverse of encode_png for the files it writes (filter type 0 only) -> uint8 [H, W, C]."""
    assert png[:8] == b'\x89PNG\r\n\x1a\n'
    i, idat, shape = 8, b'', None
    while i < len(png):
        n, = struct.unpack('>I', png[i:i + 4])
        kind, payload = png[i + 4:i + 8], png[i + 8:i + 8 + n]
        assert struct.unpack('>I', png[i + 8 + n:i + 12 + n])[0] == zlib.crc32(kind + payload) & 0xFFFFFFFF
        if kind == b'IHDR':
            w, h, depth, colour = struct.unpack('>IIBB', payload[:10])
            shape = (h, w, {0: 1, 2: 3, 6: 4}[colour])
        elif kind == b'IDAT':
            idat += payload
        i += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(shape[0], 1 + shape[1] * shape[2])
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(shape).copy()
