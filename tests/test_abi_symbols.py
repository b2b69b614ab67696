"""CPU: the C-ABI library loads and exports every symbol include/sogm_abi.h (what a plan_manager host binds) and
include/sogm_abi_debug.h (tuning, profiling, clocks, parity downloads, test hooks) declare; compute entry points fail
loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="sogm_abi.h"):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = re.findall(r"\b(sogm_[a-z_0-9]+)\s*\(", txt)
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound(pop):
    host, debug = declared_symbols(), declared_symbols("sogm_abi_debug.h")
    assert len(host) >= 18 and len(debug) >= 10 and not set(host) & set(debug)
    lib = C.CDLL(pop._abi.LIB_PATH)
    for hdr, syms in (("sogm_abi.h", host), ("sogm_abi_debug.h", debug)):
        for s in syms:
            assert hasattr(lib, s), f"{s} declared in {hdr} but not exported"
    assert set(host) | set(debug) == set(pop._abi.PROTOTYPES.keys())
    # the split: nothing a host needs to fly is in the debug header, no knob / profiler / clock /
    # parity download in the host header
    for s in host:
        assert not re.search(r"profil|tuning|_clock|traffic|history|download|force_frame", s), s
    for s in ("sogm_create", "sogm_update_world", "sogm_replan", "sogm_flight_run", "sogm_traj_allgather", "sogm_last_error"):
        assert s in host


def test_the_facade_includes_only_the_host_header():
    for f in ("sogm_facade.hpp", "sogm_reference_api.hpp"):
        txt = open(os.path.join(ROOT, "pred-occ-planner_amd", "host", f)).read()
        assert "sogm_abi_debug.h\"" not in txt.replace("include/sogm_abi_debug.h", "")
        assert not re.search(r"#include\s+\"[^\"]*sogm_abi_debug\.h\"", txt)


def test_struct_sizes_match_header(pop):
    assert pop._abi.TRAJ_RECORD_BYTES == 4 + 4 + 8 + 16 * 8 + 16 * 15 * 8
    assert pop._abi.CYLINDER_BYTES == 8 + 11 * 8
    assert C.sizeof(pop._abi.SogmSpec) == 15 * 4


def test_abi_version_and_device_count(pop):
    lib = pop.lib()
    assert lib.sogm_abi_version() == pop._abi.SOGM_ABI_VERSION == 6
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
