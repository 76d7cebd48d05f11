"""CPU: the C-ABI library builds, loads without a GPU and exports every symbol include/sac_hip.h
declares; with no GPU every constructor fails loudly (no CPU fallback anywhere in the product)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sac_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:sac|td3)_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_for_gfx950(tmp_path):
    from robosuite_benchmark_amd.build import build_library
    lib = build_library()
    assert os.path.exists(lib)
    out = "gfx950"
    if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        # --offloading extracts the bundled code objects next to the input: work on a copy in a scratch directory
        import shutil
        copy = shutil.copy(lib, tmp_path / "libsac_hip.so")
        out = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(copy)], text=True,
                                      stderr=subprocess.STDOUT, cwd=str(tmp_path))
    assert "gfx950" in out


def test_every_declared_symbol_is_exported_and_bound():
    from robosuite_benchmark_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sac_hip.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature in _lib.SYMBOLS"
    assert set(_lib.SYMBOLS) == set(names)
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T ((?:sac|td3)_[a-z0-9_]+)", nm))
    assert set(names) <= exported
    assert lib.sac_version().startswith(b"sac_hip")


def test_no_cpu_fallback_without_a_gpu():
    from robosuite_benchmark_amd import EnvReplayBuffer, FlattenMlp, SACTrainer, TanhGaussianPolicy, _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        EnvReplayBuffer(100, obs_dim=4, action_dim=2)
    pol = TanhGaussianPolicy([256, 256], 4, 2)
    qs = [FlattenMlp([256, 256], 1, 6) for _ in range(4)]
    with pytest.raises(RuntimeError, match="no HIP device"):
        SACTrainer(policy=pol, qf1=qs[0], qf2=qs[1], target_qf1=qs[2], target_qf2=qs[3], batch_size=64)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "robosuite_benchmark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f == "build.py", f
