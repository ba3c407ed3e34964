"""PLY io without plyfile (reference: vgtk/vgtk/pc/io.py:L6-60)."""
import numpy as np

_TYPES = {'char': 'i1', 'uchar': 'u1', 'short': 'i2', 'ushort': 'u2', 'int': 'i4', 'uint': 'u4',
          'float': 'f4', 'double': 'f8', 'int8': 'i1', 'uint8': 'u1', 'int16': 'i2', 'uint16': 'u2',
          'int32': 'i4', 'uint32': 'u4', 'float32': 'f4', 'float64': 'f8'}


def _read_vertices(path):
    with open(path, 'rb') as f:
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            tok = f.readline().decode('ascii').split()
            if not tok:
                continue
            if tok[0] == 'end_header':
                break
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                props.append((tok[2], _TYPES[tok[1]]))
        if fmt == 'ascii':
            rows = np.array([f.readline().split()[:len(props)] for _ in range(count)], dtype=np.float64)
            return {name: rows[:, i] for i, (name, _) in enumerate(props)}
        dt = np.dtype([(n, ('<' if fmt.endswith('little_endian') else '>') + t) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt)
        return {n: data[n] for n, _ in props}


def load_ply(file_name, with_faces=False, with_color=False, with_normal=False):
    """-> points [n,3] (vertex x,y,z); faces are not supported (unused on the conv path)."""
    if with_faces:
        raise NotImplementedError('load_ply(with_faces=True)')
    v = _read_vertices(file_name)
    points = np.vstack([v['x'], v['y'], v['z']]).T
    ret = [points]
    if with_color:
        ret.append(np.vstack([v['red'], v['green'], v['blue']]).T)
    pc = ret[0] if len(ret) == 1 else ret
    if with_normal:
        return pc, np.vstack([v['nx'], v['ny'], v['nz']]).T
    return pc


def save_ply(filepath, color_pc, c=None, use_color=False, use_normal=False, verbose=False):
    """ASCII PLY writer (points [n,3] or [n,6] with colour/normal columns)."""
    pts = np.asarray(color_pc)
    n = pts.shape[0]
    colour = use_color or c is not None
    with open(filepath, 'w') as f:
        f.write('ply\nformat ascii 1.0\nelement vertex %d\n' % n)
        f.write('property float x\nproperty float y\nproperty float z\n')
        if use_normal:
            f.write('property float nx\nproperty float ny\nproperty float nz\n')
        if colour:
            f.write('property uchar red\nproperty uchar green\nproperty uchar blue\n')
        f.write('end_header\n')
        rgb = {'r': (255, 0, 0), 'g': (0, 255, 0), 'b': (0, 0, 255)}.get(c, (255, 255, 255)) if isinstance(c, str) else None
        for i in range(n):
            line = '%f %f %f' % tuple(pts[i, :3])
            if use_normal:
                line += ' %f %f %f' % tuple(pts[i, 3:6])
            if colour:
                col = rgb if rgb is not None else tuple(int(v) for v in pts[i, 3:6])
                line += ' %d %d %d' % col
            f.write(line + '\n')
