"""The kernels' arithmetic shortcut (rcp_exact: v_rcp_f32 + one fused Newton step in place of the 11-instruction IEEE division)
must equal `1.0f / x` bit for bit for EVERY f32 — swept exhaustively on the device through the C ABI (mi_selftest)."""
import pytest

pytestmark = pytest.mark.gpu


def test_rcp_exact_equals_ieee_division_for_all_f32(gpu_ctx):
    bad, checked = gpu_ctx.selftest()
    assert checked == 1 << 32
    assert bad == 0, f"{bad} f32 inputs where the short reciprocal differs from the IEEE division"
