"""utils/summary.py: the TensorBoard event files the reference's trainers write through tf.summary (reference
models/wgancls/trainer.py:20-47) — framing, checksums, protocol-buffer encoding and the three summary kinds, checked against
published known answers (CRC-32C check value, TFRecord mask, protobuf wire examples) and by reading the file back."""
import struct

import numpy as np
import pytest

import t2i_amd  # noqa: F401
from t2i_amd.utils import summary as S


def test_crc32c_known_answers():
    assert S.crc32c(b'123456789') == 0xE3069283            # the CRC-32C (Castagnoli) check value
    assert S.crc32c(b'') == 0
    assert S.crc32c(bytes(32)) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert S.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of ones
    c = S.crc32c(b'abc')
    assert S.masked_crc32c(b'abc') == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def test_wire_format_known_answers():
    assert S._varint(1) == b'\x01' and S._varint(300) == b'\xac\x02'          # protobuf encoding guide's examples
    assert S._f_varint(1, 150) == b'\x08\x96\x01'
    assert S._f_bytes(2, 'testing') == b'\x12\x07testing'
    assert S._f_float(2, 1.0) == b'\x15' + struct.pack('<f', 1.0)
    assert S._parse(S._f_varint(1, 150) + S._f_bytes(2, 'testing')) == [(1, 0, 150), (2, 2, b'testing')]
    assert S._varint(-1) == b'\xff' * 9 + b'\x01'                              # negative int64: ten bytes


def test_event_file_round_trip(tmp_path):
    w = S.FileWriter(str(tmp_path / 'logs'))
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (5, 8, 6, 3)).astype(np.float32)
    z = rng.standard_normal((4, 100))
    w.add_summary([S.scalar('D_loss', -3.25), S.image('x', x), S.histogram('z', z)], 10)
    w.add_summary(S.scalar('kt', 0.5), 20)
    w.close()
    ev = S.read_events(w.path)
    assert ev[0]['file_version'] == 'brain.Event:2' and 'step' not in ev[0]
    assert ev[1]['step'] == 10 and ev[2]['step'] == 20 and ev[2]['values'] == [{'tag': 'kt', 'simple_value': 0.5}]
    v = ev[1]['values']
    assert [r['tag'] for r in v] == ['D_loss', 'x/image/0', 'x/image/1', 'x/image/2', 'z'] and v[0]['simple_value'] == -3.25
    im = v[1]['image']
    assert (im['height'], im['width'], im['colorspace']) == (8, 6, 3)
    got = S.decode_png(im['png'])
    assert np.array_equal(got, S.normalize_image(x[0]))
    h = v[4]['histo']
    assert h['num'] == 400 and abs(h['sum'] - z.sum()) < 1e-9 and abs(h['sum_squares'] - (z * z).sum()) < 1e-9
    assert h['min'] == z.min() and h['max'] == z.max() and sum(h['bucket']) == 400 and len(h['bucket']) == len(h['bucket_limit'])
    # a corrupted byte is caught by the frame checksum
    raw = bytearray(open(w.path, 'rb').read())
    raw[40] ^= 1
    bad = tmp_path / 'bad'
    bad.write_bytes(bytes(raw))
    with pytest.raises(ValueError):
        S.read_events(str(bad))


def test_image_normalisation_rule():
    """tf.summary.image on floats, per image: non-negative -> the largest value maps to 255; otherwise 0 -> 128 and the largest
    magnitude to 128 +- 127."""
    pos = np.array([[[0.0], [0.5]], [[1.0], [2.0]]], np.float32)
    assert S.normalize_image(pos)[..., 0].tolist() == [[0, 63], [127, 255]]
    neg = np.array([[[-1.0], [0.0]], [[0.5], [1.0]]], np.float32)
    assert S.normalize_image(neg)[..., 0].tolist() == [[1, 128], [191, 255]]
    assert S.normalize_image(np.zeros((2, 2, 1), np.float32)).max() == 0
    u8 = np.arange(12, dtype=np.uint8).reshape(2, 2, 3)
    assert S.normalize_image(u8) is u8
    one = S.image('t', np.zeros((2, 4, 4, 1), np.float32), max_outputs=1)
    assert b't/image' in one and b't/image/0' not in one


def test_histogram_buckets():
    lim = S.default_bucket_limits()
    assert lim[len(lim) // 2] == 0.0 and lim[-1] == np.finfo(np.float64).max and lim[0] == -lim[-1]
    assert lim[len(lim) // 2 + 1] == 1e-12 and abs(lim[len(lim) // 2 + 2] / 1e-12 - 1.1) < 1e-12 and len(lim) == 2 * 775 + 1
    ev = S._parse(S._parse(S.histogram('h', [0.0, 1.0, 1.0, -2.0]))[0][2])
    h = {f: v for f, _, v in S._parse(ev[1][2])}
    limits = struct.unpack('<%dd' % (len(h[6]) // 8), h[6])
    counts = struct.unpack('<%dd' % (len(h[7]) // 8), h[7])
    # every value lies below its bucket's limit and at or above the previous limit; empty runs are collapsed to one entry
    assert sum(counts) == 4 and limits == tuple(sorted(limits)) and limits[-1] == lim[-1]
    filled = [(l, c) for l, c in zip(limits, counts) if c > 0]
    assert [c for _, c in filled] == [1.0, 1.0, 2.0]
    assert filled[0][0] > -2.0 and filled[1][0] == 1e-12 and filled[2][0] > 1.0 and filled[2][0] / 1.1 <= 1.0
