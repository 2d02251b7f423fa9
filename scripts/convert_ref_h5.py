"""Converts the reference's HDF5 test fixtures (src/tests/data/*.h5) into small .npz files under
tests/golden/ (data only: inputs and expected outputs, no reference source).

The image has no h5py, so this is a minimal reader of the classic HDF5 layout those files use:
superblock v0, symbol-table groups, v1 object headers, contiguous little-endian IEEE datasets.
Run in the build container (needs /root/reference):  python scripts/convert_ref_h5.py
"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/src/tests/data'


class H5:
    def __init__(self, path):
        self.d = open(path, 'rb').read()
        assert self.d[:8] == b'\x89HDF\r\n\x1a\n'
        assert self.d[8] == 0, 'superblock version'
        so, sl = self.d[13], self.d[14]
        assert so == 8 and sl == 8
        # root symbol table entry starts after: 8 sig + 8 versions/sizes + 2+2 K + 4 flags + 4*8 addresses
        off = 8 + 8 + 4 + 4 + 32
        self.root = self._ste(off)

    def _ste(self, off):
        name_off, ohdr, cache = struct.unpack_from('<QQI', self.d, off)
        scratch = self.d[off + 24: off + 40]
        return dict(name_off=name_off, ohdr=ohdr, cache=cache, scratch=scratch)

    def _heap_name(self, heap_addr, name_off):
        assert self.d[heap_addr:heap_addr + 4] == b'HEAP'
        data_addr = struct.unpack_from('<Q', self.d, heap_addr + 24)[0]
        s = data_addr + name_off
        e = self.d.index(b'\x00', s)
        return self.d[s:e].decode()

    def _btree_entries(self, addr, heap):
        assert self.d[addr:addr + 4] == b'TREE', addr
        ntype, level, nent = struct.unpack_from('<BBH', self.d, addr + 4)
        out = []
        p = addr + 8 + 16   # siblings
        for i in range(nent):
            p += 8          # key
            child = struct.unpack_from('<Q', self.d, p)[0]
            p += 8
            if level > 0:
                out += self._btree_entries(child, heap)
            else:
                assert self.d[child:child + 4] == b'SNOD'
                nsym = struct.unpack_from('<H', self.d, child + 6)[0]
                for k in range(nsym):
                    e = self._ste(child + 8 + 40 * k)
                    out.append((self._heap_name(heap, e['name_off']), e))
        return out

    def _messages(self, ohdr):
        ver, _, nmsg, _, hsize = struct.unpack_from('<BBHII', self.d, ohdr)
        assert ver == 1
        msgs = []
        blocks = [(ohdr + 16, hsize)]
        while blocks and len(msgs) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(msgs) < nmsg:
                mtype, msize, mflags = struct.unpack_from('<HHB', self.d, p)
                body = self.d[p + 8: p + 8 + msize]
                if mtype == 0x10:
                    coff, clen = struct.unpack_from('<QQ', body, 0)
                    blocks.append((coff, clen))
                msgs.append((mtype, body))
                p += 8 + msize
        return msgs

    def items(self, entry=None):
        entry = entry or self.root
        btree, heap = struct.unpack_from('<QQ', entry['scratch'], 0) if entry['cache'] == 1 else (None, None)
        if btree is None:
            for mtype, body in self._messages(entry['ohdr']):
                if mtype == 0x11:
                    btree, heap = struct.unpack_from('<QQ', body, 0)
        return self._btree_entries(btree, heap)

    def dataset(self, entry):
        dims = dtype = addr = size = None
        for mtype, body in self._messages(entry['ohdr']):
            if mtype == 0x1:
                ver, rank = body[0], body[1]
                off = 8 if ver == 1 else 4
                dims = struct.unpack_from('<' + 'Q' * rank, body, off)
            elif mtype == 0x3:
                cls = body[0] & 0x0f
                sz = struct.unpack_from('<I', body, 4)[0]
                assert cls in (0, 1), cls
                dtype = {(1, 8): '<f8', (1, 4): '<f4', (0, 4): '<i4', (0, 8): '<i8'}[(cls, sz)]
            elif mtype == 0x8:
                ver = body[0]
                if ver == 3:
                    assert body[1] == 1, 'only contiguous layout'
                    addr, size = struct.unpack_from('<QQ', body, 2)
                else:
                    rank, cls = body[1], body[2]
                    assert cls == 1
                    addr = struct.unpack_from('<Q', body, 8)[0]
        n = int(np.prod(dims))
        arr = np.frombuffer(self.d, dtype=dtype, count=n, offset=addr).reshape(dims)
        return np.array(arr, dtype=np.float64 if 'f' in dtype else np.int64)

    def read_all(self):
        return {name: self.dataset(e) for name, e in self.items()}


def main():
    out = os.path.join(ROOT, 'tests', 'golden')
    for fn in ('test_error_feature_quadric.h5', 'test_error_bbox_quadric.h5'):
        data = H5(os.path.join(REF, fn)).read_all()
        np.savez_compressed(os.path.join(out, 'ref_' + fn[:-3] + '.npz'), **data)
        print(fn, {k: v.shape for k, v in data.items()})
    frames = {}
    for i in range(47):
        data = H5(os.path.join(REF, 'one_car', f'frame_{i}.h5')).read_all()
        for k, v in data.items():
            frames.setdefault(k, []).append(v)
    np.savez_compressed(os.path.join(out, 'ref_one_car.npz'), **{k: np.stack(v) for k, v in frames.items()})
    print('one_car', {k: np.stack(v).shape for k, v in frames.items()})


if __name__ == '__main__':
    main()
