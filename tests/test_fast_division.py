"""CPU: the three-instruction fp32 division by the voxel resolution (GridGeom::div_res, csrc/sogm_device.hpp) —
q0 = a * RN(1 / res), e = fma(-res, q0, a), q = fma(e, RN(1 / res), q0) — equals the correctly rounded a / res, the
reference's `(pos + range) / resolution_` (plan_env/include/plan_env/map.h:169-174), for res = 0.15f and |a| < 64.
The kernels use it only for that resolution (VOXEL_RESOLUTION, a compile-time constant of the reference) and that range.
Checked here: every 5th float of [2^-20, 64) and every float within 64 ulps of a multiple of res (where the truncation
to a voxel index could flip); the FULL range was checked once when the sequence was introduced (1.1e9 values, 0
mismatches — profiles/EXPERIMENTS.md round 5).  The fused multiply-adds are emulated in 80-bit extended precision: the
residual e is exact in float32 (asserted), the final sum is rounded once more than a hardware FMA would — a mismatch
would show as a failure here, never hide one.  The GPU side is held to the oracle's true division by every cell-exact
parity test."""
import numpy as np

D = np.float32(0.15)
INV = np.float32(1.0 / np.float64(D))
LD = np.longdouble


def _div_res(a):
    q0 = a * INV
    e_x = a.astype(LD) - LD(D) * q0.astype(LD)
    e = e_x.astype(np.float32)
    assert np.all(e.astype(LD) == e_x), "fma(-res, q0, a) must be exact in float32"
    return (q0.astype(LD) + e.astype(LD) * LD(INV)).astype(np.float32)


def test_reciprocal_constant_is_the_one_the_library_checks_for():
    assert float(INV) == 6.666666507720947 and INV == np.float32(1) / D


def test_sampled_range_is_correctly_rounded():
    lo, hi = int(np.float32(2.0 ** -20).view(np.uint32)), int(np.float32(64.0).view(np.uint32))
    n = 0
    for start in range(lo, hi, 1 << 25):
        a = np.arange(start, min(start + (1 << 25), hi), 5, dtype=np.uint32).view(np.float32)
        q = _div_res(a)
        assert np.array_equal(q, a / D)
        assert np.array_equal(_div_res(-a), (-a) / D)
        n += len(a)
    assert n > 4e7


def test_every_float_near_a_voxel_boundary_truncates_like_the_true_division():
    k = np.arange(1, 427, dtype=np.float64)      # voxel boundaries up to 64 m
    centre = (k * np.float64(D)).astype(np.float32).view(np.uint32).astype(np.int64)
    bits = (centre[:, None] + np.arange(-64, 65)[None, :]).astype(np.uint32).ravel()
    a = bits.view(np.float32)
    q, t = _div_res(a), a / D
    assert np.array_equal(q, t)
    assert np.array_equal(q.astype(np.int32), t.astype(np.int32))
    # values below 2^-20 truncate to voxel 0 either way
    tiny = np.float32([0.0, 1e-30, 1e-12, 2.0 ** -21, -1e-12])
    assert np.all(_div_res(tiny).astype(np.int32) == 0) and np.all((tiny / D).astype(np.int32) == 0)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_every_float_on_the_device():
    """EVERY float of [2^-20, 64) and its negative through GridGeom::div_res on the device (the hardware's own fused
    multiply-adds, no emulation) against the device's IEEE division: 0 mismatches (sogm_debug_div_check; 4.4e8 operands)."""
    import ctypes as C
    import importlib
    pop = importlib.import_module("pred-occ-planner_amd")
    sogm = importlib.import_module("pred-occ-planner_amd.sogm")
    lib = pop.lib()
    lib.sogm_debug_div_check.restype = C.c_int
    lib.sogm_debug_div_check.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    m = sogm.SogmMap(pop.config.make_spec("parity"), 1)
    out = (C.c_uint64 * 3)()
    assert lib.sogm_debug_div_check(m.ctx, 2.0 ** -20, 64.0, out) == 0
    assert out[2] == 1, "the parity grid's resolution (0.15f) must take the fast sequence"
    assert out[0] == 0, f"{out[0]} mismatches, first at bit pattern {out[1]:#x}"
    # ... and the range it is NOT used for stays the true division by construction
    assert lib.sogm_debug_div_check(m.ctx, 64.0, 1.0e6, out) == 0 and out[0] == 0
    m.close()
