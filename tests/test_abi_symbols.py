"""CPU: the C-ABI library loads and exports every symbol include/sogm_abi.h declares; compute
entry points fail loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sogm_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(sogm_[a-z_0-9]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound(pop):
    syms = declared_symbols()
    assert len(syms) >= 18
    lib = C.CDLL(pop._abi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in sogm_abi.h but not exported"
    assert set(syms) == set(pop._abi.PROTOTYPES.keys())


def test_struct_sizes_match_header(pop):
    assert pop._abi.TRAJ_RECORD_BYTES == 4 + 4 + 8 + 16 * 8 + 16 * 15 * 8
    assert pop._abi.CYLINDER_BYTES == 8 + 11 * 8
    assert C.sizeof(pop._abi.SogmSpec) == 15 * 4


def test_abi_version_and_device_count(pop):
    lib = pop.lib()
    assert lib.sogm_abi_version() == pop._abi.SOGM_ABI_VERSION == 5
    assert lib.sogm_device_count() >= 0


def test_create_without_gpu_fails_loudly(pop):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = pop.lib()
    ctx = C.c_void_p()
    spec = pop.config.make_spec()
    rc = lib.sogm_create(C.byref(spec), 1, 0, C.byref(ctx))
    assert rc == pop._abi.SOGM_ERR_NO_DEVICE and not ctx.value
    with pytest.raises(pop.SogmError):
        pop._abi.check(rc, "sogm_create")


def test_invalid_arguments_rejected(pop):
    lib = pop.lib()
    ctx = C.c_void_p()
    spec = pop.config.make_spec()
    spec.T = 0
    assert lib.sogm_create(C.byref(spec), 1, 0, C.byref(ctx)) == pop._abi.SOGM_ERR_INVALID_ARG
    spec = pop.config.make_spec()
    assert lib.sogm_create(C.byref(spec), 0, 0, C.byref(ctx)) == pop._abi.SOGM_ERR_INVALID_ARG


def test_missing_extension_raises(pop, tmp_path):
    with pytest.raises(pop.SogmError):
        pop.load_library(str(tmp_path / "nope.so"))
