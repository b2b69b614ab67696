"""sogm_set_tuning / sogm_get_tuning: the library's knobs are per-context values with documented defaults, not
environment variables (the library reads three environment switches: SOGM_SPARSE_RESET, SOGM_FLOW, SOGM_RCCL_LIB)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_library_reads_three_environment_switches_only():
    names = set()
    for f in glob.glob(os.path.join(ROOT, "pred-occ-planner_amd", "csrc", "*.h*")):
        names |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', open(f).read()))
    assert names == {"SOGM_SPARSE_RESET", "SOGM_FLOW", "SOGM_RCCL_LIB"}, names
    # ... and keeps no function-static copies of tuning values
    for f in glob.glob(os.path.join(ROOT, "pred-occ-planner_amd", "csrc", "sogm_map.hip")) + \
            glob.glob(os.path.join(ROOT, "pred-occ-planner_amd", "csrc", "sogm_planner.hip")):
        assert not re.findall(r"static (const )?(int|double)\s+\w+\s*=\s*-1", open(f).read()), f


def test_every_key_the_header_documents_is_a_key_of_the_library(pop):
    hdr = open(os.path.join(ROOT, "include", "sogm_abi_debug.h")).read()
    doc = hdr[hdr.index("/* Tuning knobs."):hdr.index("int         sogm_set_tuning")]
    documented = set(re.findall(r'"([a-z_]+)"', doc))
    lib = pop.lib()
    keys, i = set(), 0
    while True:
        k = lib.sogm_tuning_key(i)
        if k is None:
            break
        keys.add(k.decode())
        i += 1
    assert keys == documented, (keys ^ documented)


@pytest.mark.gpu
def test_set_and_get_tuning(pop):
    import importlib
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    m = sogm.SogmMap(pop.config.make_spec("parity"), 2)
    assert m.get_tuning("reset_wgs") == 32 and m.get_tuning("groups") == 2 and m.get_tuning("clear_head_gb") == 1.0e9
    m.set_tuning("reset_wgs", 16)
    assert m.get_tuning("reset_wgs") == 16
    with pytest.raises(pop._abi.SogmError):
        m.set_tuning("no_such_key", 1)
    # values become launch dimensions and ticket counts: non-finite and out-of-range ones are refused
    for bad in (float("inf"), float("-inf"), float("nan"), -1.0, 1e12):
        with pytest.raises(pop._abi.SogmError):
            m.set_tuning("prestamp_wgs", bad)
    assert m.get_tuning("prestamp_wgs") == 0
    m.close()


def test_malformed_tuning_environment_item_is_named(pop, monkeypatch):
    import importlib
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    src = open(sogm.__file__).read()
    assert "is not key=number" in src  # (the constructor needs a GPU; the message is what the fix added)
