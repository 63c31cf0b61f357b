"""Measurement support of bench.py's `valu_frac` figures: aomhip_valu_issue_probe runs every opcode class and reports a plausible rate."""
import pytest

import aom_av1_psy_amd as pkg

pytestmark = pytest.mark.gpu
capi = pkg.capi


def test_valu_issue_probe_reports_every_opcode_class():
    ctx = capi.Context(0)
    names = capi.valu_issue_probe_names()
    assert len(names) >= 25 and names[0] == "v_add_u32" and len(set(names)) == len(names)
    rates = {}
    for op, nm in enumerate(names):
        r = ctx.valu_issue_probe(op, 8, 300)
        assert r["launch_ms"] > 0 and r["waves_per_simd"] == 8 and r["compute_units"] >= 64
        assert 1e7 < r["wave_insts_per_s_per_simd"] < 3e9, (nm, r)     # between 1/60 and 1.25 instructions per clock at ~2.4 GHz
        assert r["memtime_ticks_per_wave_inst"] > 0 and 2e7 < r["memtime_hz"] < 4e9
        rates[nm] = r["wave_insts_per_s_per_simd"]
    # relations measured on MI355X (profiles/r05_valu_issue.md): most integer / packed / dot / SAD opcodes retire one wave64 instruction per
    # ~4 clocks per SIMD, v_add_u32 and the fp32 add / fma one per ~2; transcendentals and fp64 reciprocal are slower
    assert rates["v_add_u32"] > 1.3 * rates["v_sad_u8"] and rates["v_fma_f32"] > 1.3 * rates["v_sad_u8"]
    assert rates["v_exp_f32"] < 0.7 * rates["v_sad_u8"] and rates["v_rcp_f64"] < 0.5 * rates["v_sad_u8"]
    assert 0.8 < rates["v_mul_u32_u24"] / rates["v_mad_i32_i24"] < 1.25
    # one wavefront per SIMD cannot issue faster than eight
    one = ctx.valu_issue_probe(0, 1, 300)
    assert one["wave_insts_per_s_per_simd"] <= 1.05 * ctx.valu_issue_probe(0, 8, 300)["wave_insts_per_s_per_simd"]
    ctx.close()


def test_valu_issue_probe_rejects_bad_arguments():
    ctx = capi.Context(0)
    r = capi.ValuProbeResult()
    import ctypes as C
    assert capi.lib.aomhip_valu_issue_probe(ctx.h, -1, 8, 10, C.byref(r)) == capi.ERR_INVALID
    assert capi.lib.aomhip_valu_issue_probe(ctx.h, len(capi.valu_issue_probe_names()), 8, 10, C.byref(r)) == capi.ERR_INVALID
    assert capi.lib.aomhip_valu_issue_probe(ctx.h, 0, 3, 10, C.byref(r)) == capi.ERR_INVALID
    assert capi.lib.aomhip_valu_issue_probe(ctx.h, 0, 8, 0, C.byref(r)) == capi.ERR_INVALID
    ctx.close()
