"""CPU-side checks of the drop-in boundary: libcge_amd.so builds, loads and exports exactly the
entry points include/cge_amd.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "cge_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cge_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_full_per_env_surface():
    names = _declared()
    for env in ["snake", "crypto", "traffic"]:
        for fn in ["create", "destroy", "seed", "reset", "step", "rollout", "info", "state_bytes", "get_state",
                   "set_state", "last_error", "device_bytes"]:
            assert f"cge_{env}_{fn}" in names, (env, fn)
    assert "cge_snake_error_count" in names
    for fn in ["create", "destroy", "seed", "reset", "step", "rollout", "info", "info64", "last_error", "device_bytes"]:
        assert f"cge_parking_{fn}" in names, fn
    for env in ["climate", "fleet", "manufacturing", "hospital"]:
        for fn in ["create", "destroy", "seed", "reset", "step", "rollout", "info", "last_error", "device_bytes"]:
            assert f"cge_{env}_{fn}" in names, (env, fn)
    for env in ["parking", "climate", "fleet", "manufacturing", "hospital"]:      # whole-handle checkpoint / resume
        for fn in ["snapshot_bytes", "snapshot_get", "snapshot_set"]:
            assert f"cge_{env}_{fn}" in names, (env, fn)


def test_library_exports_every_declared_symbol():
    from custom_gymnasium_environments_amd import _native, build
    build.build_native()
    L = ctypes.CDLL(_native.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, missing
    # and the python binding table covers exactly the header
    assert sorted(_native.SIGNATURES) == _declared()


def test_host_only_entry_points_work_without_a_gpu(oracle):
    from custom_gymnasium_environments_amd import _native
    L = _native.lib()
    assert b"gfx950" in L.cge_version()
    for args in [(123, 0, 0, 4, 0), (123, 1048575, 999, 4, 0), (7, 77, 12345, 3, 8), (2**63, 5, 6, 5, 0)]:
        assert L.cge_hash_action(*args) == oracle.hash_action(*args)


def test_no_cpu_fallback_and_loud_failure():
    import torch
    import custom_gymnasium_environments_amd as cge
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cge.NativeLibraryError):
        cge.SnakeVectorEnv(8, grid_size=10)
    with pytest.raises(cge.NativeLibraryError):
        cge.SnakeVectorEnv(8, grid_size=10, device="cpu")
    for name in ["Crypto", "Traffic", "Parking", "Climate", "Fleet", "Manufacturing", "Hospital"]:     # every env type: no silent CPU path
        with pytest.raises(cge.NativeLibraryError):
            getattr(cge, name + "VectorEnv")(8)
        with pytest.raises(cge.NativeLibraryError):
            getattr(cge, name + "VectorEnv")(8, device="cpu")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "custom_gymnasium_environments_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "orc_" not in text, f


def test_registry_covers_the_reference_ids():
    """The ids the reference registers with gymnasium / RLlib (file:line in custom_gymnasium_environments_amd/registry.py) resolve
    to the batched classes; an unknown id is an error, not a fallback."""
    import custom_gymnasium_environments_amd as cge
    assert cge.registered_ids() == sorted(["snake_env_classic-v0", "CryptoTrading-v0", "TrafficManagement-v0", "SmartParkingEnv-v0",
                                           "SmartClimateEnv-v0", "FleetManagement-v0", "HospitalManagement-v0", "SmartManufacturing-v0"])
    with pytest.raises(ValueError):
        cge.make_vec("CartPole-v1", 4)
    for name in ["reset", "step", "close", "call", "get_attr", "set_attr", "render", "unwrapped"]:
        assert hasattr(cge.NumpyVectorEnv, name), name
