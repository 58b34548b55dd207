"""INTEGRATION.md section 2 shows the ctypes binding a maintainer of the reference would add (`utils/samble_hip.py`).
The snippet is executed here as it stands in the document -- only the library's file name is pointed at the in-tree
build -- and its `knn` is held against the oracle: a documented binding that does not run is worse than none."""
import os
import re

import pytest
import torch

from oracle import torch_oracle as O
from samble_amd import synth
from tests.util import set_agreement

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "utils/samble_hip.py" in b)
    lib = os.path.join(ROOT, "samble_amd", "libsamble_hip.so")
    assert 'ctypes.CDLL("libsamble_hip.so")' in code
    return code.replace('ctypes.CDLL("libsamble_hip.so")', f'ctypes.CDLL("{lib}")')


def test_documented_binding_names_the_current_abi():
    from samble_amd import _lib
    code = _snippet()
    assert f"samble_abi_version() == {_lib.ABI_VERSION}" in code
    # the two entry points it binds, with the header's arities
    for name, args in (("samble_knn_workspace_bytes", 6), ("samble_knn_f32", 15)):
        m = re.search(name + r"\.argtypes = \[(.*?)\]", code, flags=re.S)
        assert m and len([a for a in m.group(1).replace("\n", " ").split(",") if a.strip()]) == args, name
        assert len(_lib._SIGNATURES[name][1]) == args, name


@pytest.mark.gpu
def test_documented_binding_runs_and_matches_the_oracle():
    ns = {}
    exec(compile(_snippet(), "INTEGRATION.md:utils/samble_hip.py", "exec"), ns)   # noqa: S102 (our own document)
    B, N, C, K = 2, 300, 128, 32
    pts = torch.from_numpy(synth.features(B, C, N, 9)).permute(0, 2, 1).contiguous()          # (B,N,C) as the reference holds them
    d, i = ns["knn"](pts.cuda(), pts.cuda(), K)
    rd, ri = O.knn(pts, pts, K)
    assert i.dtype == torch.int64 and i.shape == (B, N, K) and d.shape == (B, N, K)
    assert set_agreement(i.cpu(), ri) >= 0.9995
    torch.testing.assert_close(d.cpu()[:, :, 1:], rd[:, :, 1:], rtol=1e-3, atol=1e-3)
    with pytest.raises(RuntimeError, match="samble_knn_f32"):
        ns["knn"](pts.cuda(), pts.cuda()[:, :8], 32)   # more neighbours than keys: the library's own message
