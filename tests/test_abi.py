"""The C-ABI library loads and exports every symbol include/e3dgs_hip.h declares (no GPU needed)."""
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "e3dgs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(e3dgs_[a-z0-9_]+)\s*\(", text)) - {"e3dgs_alloc_fn"})


def test_header_symbols_exported(hip_lib):
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(hip_lib, s), f"{s} declared in include/e3dgs_hip.h but not exported"


def test_loader_symbol_list_matches_header():
    from event_3dgs_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_abi_version_and_scratch_sizes(hip_lib):
    assert hip_lib.e3dgs_abi_version() == 17
    assert hip_lib.e3dgs_knn_scratch_bytes(1000) > 1000 * 16
    assert hip_lib.e3dgs_event_loss_scratch_bytes(1920, 1080) >= 5 * 8


def test_product_never_imports_oracle():
    """The product packages must not route through the oracle (test infrastructure)."""
    bad = []
    for pkg in ("event_3dgs_amd", "diff_gaussian_rasterization", "simple_knn"):
        for dp, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "gs_oracle" in src:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from event_3dgs_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.HipLibraryError):
        _lib.lib()


def test_cpu_tensors_are_rejected():
    """No silent CPU fallback: the op raises on CPU tensors."""
    import pytest
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    r = GaussianRasterizer(rs)
    with pytest.raises(RuntimeError):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1),
          colors_precomp=torch.ones(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.ones(4, 1))
