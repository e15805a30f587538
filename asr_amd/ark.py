"""Kaldi scp/ark feature IO for `DataReader` (the input format of evaluate.py).

Counterpart of the reference's `third_party/feat_convert/io/ark.py` (ArkReader :29-205,
ArkWriter :209-236): same scp line format `uttid path:offset`, same binary headers
(`\\0B` + token `FM`/`DM`/`CM`/`CM2`/`FV`/`DV`), same `read_next_utt()` protocol
(returns `(uttid, matrix, looped)`).  Written from the Kaldi binary layout, vectorised with numpy.
"""
from __future__ import annotations

import struct

import numpy as np


class ArkReader(object):
    def __init__(self, scp_path):
        self.scp_position = 0
        self.utt_ids = []
        self.scp_data = []
        with open(scp_path, 'r') as f:
            for line in f:
                line = line.rstrip('\n')
                if not line.strip():
                    continue
                utt_id, path_pos = line.split(' ', 1)
                path, pos = path_pos.rsplit(':', 1)
                self.utt_ids.append(utt_id)
                self.scp_data.append((path, int(pos)))
        self._index = {u: i for i, u in enumerate(self.utt_ids)}

    @staticmethod
    def _read_token(f):
        tok = b''
        while True:
            ch = f.read(1)
            if ch in (b' ', b''):
                return tok.decode('ascii')
            tok += ch

    @staticmethod
    def _dims(f):
        _, rows = struct.unpack('<bi', f.read(5))
        _, cols = struct.unpack('<bi', f.read(5))
        return rows, cols

    def read_utt_data(self, index):
        path, pos = self.scp_data[index]
        with open(path, 'rb') as f:
            f.seek(pos, 0)
            f.read(1)                                  # the '\0' before 'B'
            header = self._read_token(f)
            if not header.startswith('B'):
                raise IOError('%s: not a binary ark entry at offset %d' % (path, pos))
            if header == 'BFM':
                rows, cols = self._dims(f)
                return np.frombuffer(f.read(rows * cols * 4), dtype='<f4').reshape(rows, cols).astype(np.float32)
            if header == 'BDM':
                rows, cols = self._dims(f)
                return np.frombuffer(f.read(rows * cols * 8), dtype='<f8').reshape(rows, cols).astype(np.float32)
            if header == 'BCM':                        # 8-bit compressed, per-column percentiles
                vmin, vrange, rows, cols = struct.unpack('<ffii', f.read(16))
                pct = np.frombuffer(f.read(cols * 8), dtype='<u2').reshape(cols, 4).astype(np.float64)
                p = vmin + vrange * pct / 65535.0      # p0, p25, p75, p100 per column
                c = np.frombuffer(f.read(rows * cols), dtype=np.uint8).reshape(cols, rows).astype(np.float64)
                p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
                lo = p0 + (p25 - p0) / 64.0 * c
                mid = p25 + (p75 - p25) / 128.0 * (c - 64)
                hi = p75 + (p100 - p75) / 63.0 * (c - 192)
                out = np.where(c <= 64, lo, np.where(c <= 192, mid, hi))
                return np.ascontiguousarray(out.T).astype(np.float32)
            if header == 'BCM2':                       # 16-bit compressed
                vmin, vrange, rows, cols = struct.unpack('<ffii', f.read(16))
                raw = np.frombuffer(f.read(rows * cols * 2), dtype='<i2').astype(np.float32)
                return (raw * np.float32(vrange) / np.float32(65535.0) + np.float32(vmin)).reshape(rows, cols)
            if header == 'BFV':
                _, size = struct.unpack('<bi', f.read(5))
                return np.frombuffer(f.read(size * 4), dtype='<f4').astype(np.float32)
            if header == 'BDV':
                _, size = struct.unpack('<bi', f.read(5))
                return np.frombuffer(f.read(size * 8), dtype='<f8').astype(np.float32)
            raise IOError('%s: unsupported ark header %r' % (path, header))

    def read_next_utt(self):
        if not self.scp_data:
            return None, None, True
        looped = self.scp_position >= len(self.scp_data)
        if looped:
            self.scp_position = 0
        self.scp_position += 1
        i = self.scp_position - 1
        return self.utt_ids[i], self.read_utt_data(i), looped

    def read_utt(self, utt_id):
        return self.read_utt_data(self._index[utt_id])

    def __len__(self):
        return len(self.utt_ids)


class ArkWriter(object):
    """Append float32 matrices as `BFM ` entries and write the matching scp lines."""

    def __init__(self, scp_path):
        self.scp_path = scp_path
        self._scp = open(scp_path, 'w')

    def write_next_utt(self, ark_path, utt_id, utt_mat):
        mat = np.ascontiguousarray(utt_mat, dtype='<f4')
        rows, cols = mat.shape
        with open(ark_path, 'ab') as f:
            f.write(utt_id.encode('utf-8'))
            pos = f.tell()
            f.write(b'\0BFM ')
            f.write(struct.pack('<bi', 4, rows))
            f.write(struct.pack('<bi', 4, cols))
            f.write(mat.tobytes())
        self._scp.write('%s %s:%d\n' % (utt_id, ark_path, pos))

    def close(self):
        self._scp.close()
