"""The C-ABI library loads without a GPU and exports exactly what include/ngp_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "ngp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ngp_[a-zA-Z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from raw_ngp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ngp_hip.h but not exported"
    assert sorted(_lib.declared_symbols()) == names, "ctypes signatures out of sync with the header"
    _lib.load()
    assert _lib.load().ngp_abi_version() == 1


def test_header_cites_the_reference_interfaces():
    hdr = open(os.path.join(ROOT, "include", "ngp_hip.h")).read()
    for cite in ("gridencoder/src/gridencoder.h:12-16", "shencoder/src/shencoder.h:8-9",
                 "raymarching/src/raymarching.h:7-19", "raymarching.py:319-329"):
        assert cite in hdr


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "raw_ngp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libngp_oracle" not in text, f


def test_shims_fail_loudly_without_device_tensors():
    import torch
    from raw_ngp_amd import _lib
    t = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _lib.shencoder_backend.sh_encode_forward(t, torch.zeros(4, 16), 4, 3, 4, None)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        _lib.raymarching_backend.morton3D(torch.zeros(4, 3, dtype=torch.int32), 4, torch.zeros(4, dtype=torch.int32))


def test_bench_refuses_a_multi_gpu_request_it_cannot_serve():
    """bench.py --gpus 2 on a box without two GPUs (and without the one-device rehearsal switches) exits non-zero before it
    touches anything -- it never reports a one-GPU number under the two-GPU name."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a multi-GPU box serves the request")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NGP_LOCAL_DEVICE")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "{" not in out.stdout
    assert "refusing" in out.stderr
