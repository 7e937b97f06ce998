"""The render kernels replace the eleven-instruction IEEE `1.0f / x` by v_rcp_f32 + one Newton step where that is the correctly rounded quotient
(dev_common.h rcp_exact / rcp_exact_large / rcp_exact3) and keep the IEEE sequence for every other input.  Bit-exact parity rests on that claim, so it
is checked here on the device at hand, over ALL 2^32 inputs, by the kernel the library ships for this purpose."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_short_reciprocals_equal_ieee_division_for_every_float(crt):
    ctx = crt.Context(64, 64)
    out = np.zeros(4, np.uint64)
    ctx.L.crt_debug_check_reciprocals.restype = C.c_int
    ctx.L.crt_debug_check_reciprocals.argtypes = [C.c_void_p, C.c_void_p]
    assert ctx.L.crt_debug_check_reciprocals(ctx.h, out.ctypes.data) == 0
    assert int(out[0]) == 1 << 32
    assert [int(v) for v in out[1:]] == [0, 0, 0], "rcp_exact / rcp_exact_large / rcp_exact3 differ from 1.0f / x: %s" % out[1:]
