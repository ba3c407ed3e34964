"""CPU: the C-ABI shared library loads and exports every symbol include/eap_hip.h declares
(no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'eap_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(eap_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported():
    so = os.path.join(ROOT, 'equi-articulated-pose_amd', 'libeap_hip.so')
    assert os.path.exists(so), 'build first: python -c "import __graft_entry__ as g; g.build()"'
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f'declared in include/eap_hip.h but not exported: {missing}'
    assert lib.eap_abi_version() == 1


def test_package_refuses_without_library(tmp_path, monkeypatch):
    """The product path must fail loudly when the HIP extension is missing (no CPU fallback)."""
    import importlib.util
    src = os.path.join(ROOT, 'equi-articulated-pose_amd', 'vgtk', '_hip.py')
    pkg = tmp_path / 'fake' / 'vgtk'
    pkg.mkdir(parents=True)
    (pkg / '_hip.py').write_text(open(src).read())
    spec = importlib.util.spec_from_file_location('fake_hip', str(pkg / '_hip.py'))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('importing vgtk._hip without libeap_hip.so must raise ImportError')


def test_host_tensors_are_rejected():
    import torch
    import vgtk.cuda.grouping as G
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    import pytest
    with pytest.raises(RuntimeError):
        G.ball_query(torch.zeros(1, 3, 4), torch.zeros(1, 3, 4), 0.1, 2)
    conv = sptk.InterSO3PoseConv(1, 4, 1, 1, 0.1, 0.01, 4)
    with pytest.raises(RuntimeError):
        conv(zptk.SphericalPointCloudPose(torch.zeros(1, 3, 8), torch.ones(1, 1, 8, 60), None, None))


def test_zpconv_backward_workspace_covers_every_aligned_chunk():
    """The scratch size the backward asks for (host-only entry, no GPU) is what its launcher carves: eleven chunks,
    each rounded up to 256 bytes (round-2 advisor finding: the old closed form fell short by up to ~2 KB for small or
    odd shapes, e.g. b=1, np=1, ann=4, nq=1)."""
    lib = ctypes.CDLL(os.path.join(ROOT, 'equi-articulated-pose_amd', 'libeap_hip.so'))
    lib.eap_inter_zpconv_bwd_workspace.restype = ctypes.c_int64

    def up(x):
        return (x + 255) // 256 * 256

    for (b, np_, nq, na, ann, c) in ((1, 1, 1, 4, 4, 16), (1, 11, 3, 28, 32, 16), (3, 5, 7, 60, 12, 16), (2, 4096, 4096, 60, 64, 64),
                                     (65, 3, 9, 60, 4, 32)):
        ent, fl = b * np_ * ann, 64 * ((b + 63) // 64)
        need = (up(4 * fl) + up(4 * ent) + up(16 * ent) + 4 * up(4 * b * nq) + up(4 * fl) + up(4 * ent) + up(16 * ent) + up(4 * ent * na * c))
        assert lib.eap_inter_zpconv_bwd_workspace(b, np_, nq, na, ann, c) == need, (b, np_, nq, na, ann, c)
