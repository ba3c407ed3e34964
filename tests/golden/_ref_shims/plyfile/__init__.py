"""Minimal stand-in for the `plyfile` package (absent from this image).

Test infrastructure only: lets `tests/golden/make_golden.py` import the
reference's Python operator path (`/root/reference/vgtk`) in THIS container so
golden vectors can be generated.  Supports exactly what the reference calls on
the hot path: ``PlyData.read(path)['vertex'][name]`` for ASCII and
binary-little-endian PLY files (reference call site: vgtk/vgtk/pc/io.py:L6-10).
Never shipped, never imported by the product package.
"""
import numpy as np

_TYPES = {
    'char': 'i1', 'uchar': 'u1', 'short': 'i2', 'ushort': 'u2', 'int': 'i4',
    'uint': 'u4', 'float': 'f4', 'double': 'f8', 'int8': 'i1', 'uint8': 'u1',
    'int16': 'i2', 'uint16': 'u2', 'int32': 'i4', 'uint32': 'u4',
    'float32': 'f4', 'float64': 'f8',
}


def _parse_header(f):
    fmt = None
    elements = []  # (name, count, [(kind, ...)])
    while True:
        line = f.readline().decode('ascii').strip()
        if line == 'end_header':
            break
        tok = line.split()
        if not tok:
            continue
        if tok[0] == 'format':
            fmt = tok[1]
        elif tok[0] == 'element':
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == 'property':
            if tok[1] == 'list':
                elements[-1][2].append(('list', tok[2], tok[3], tok[4]))
            else:
                elements[-1][2].append(('scalar', tok[1], tok[2]))
    return fmt, elements


class _Element(dict):
    pass


class PlyData(dict):
    @staticmethod
    def read(path):
        out = PlyData()
        with open(path, 'rb') as f:
            fmt, elements = _parse_header(f)
            for name, count, props in elements:
                cols = {p[-1]: [] for p in props}
                for _ in range(count):
                    if fmt == 'ascii':
                        tok = f.readline().decode('ascii').split()
                        pos = 0
                        for p in props:
                            if p[0] == 'scalar':
                                cols[p[2]].append(np.dtype(_TYPES[p[1]]).type(tok[pos]))
                                pos += 1
                            else:
                                n = int(tok[pos]); pos += 1
                                cols[p[3]].append(np.array(tok[pos:pos + n], dtype=_TYPES[p[2]]))
                                pos += n
                    else:
                        for p in props:
                            if p[0] == 'scalar':
                                dt = np.dtype('<' + _TYPES[p[1]])
                                cols[p[2]].append(np.frombuffer(f.read(dt.itemsize), dt)[0])
                            else:
                                ct = np.dtype('<' + _TYPES[p[1]])
                                n = int(np.frombuffer(f.read(ct.itemsize), ct)[0])
                                dt = np.dtype('<' + _TYPES[p[2]])
                                cols[p[3]].append(np.frombuffer(f.read(dt.itemsize * n), dt).copy())
                el = _Element()
                for p in props:
                    key = p[-1]
                    if p[0] == 'scalar':
                        el[key] = np.array(cols[key], dtype=_TYPES[p[1]])
                    else:
                        el[key] = cols[key]
                out[name] = el
        return out


class PlyElement:  # imported by the reference, unused on the hot path
    @staticmethod
    def describe(*a, **k):
        raise NotImplementedError
