"""aomhip_vbp_8x8_stats_plane / aomhip_vbp_4x4_avg_plane (csrc/var_part.hip) against (a) fill_variance_8x8avg / compute_minmax_8x8 /
fill_variance_4x4avg interpreted (tests/golden/ref_eval_vbp.npz, directly) and (b) the oracle on whole planes."""
import numpy as np
import pytest

from test_golden_vbp import load, oracle_4x4, oracle_8x8, planes_of

pytestmark = pytest.mark.gpu


def test_device_matches_the_interpreted_functions(hip, ctx):
    z, cases = load()
    done = set()
    n8 = n4 = 0
    for c in cases:
        key = (c["kind"], c["bd"], c["pw"], c["ph"], c.get("border_offset"))
        src, dst = planes_of(z, c["bd"])
        S = src.shape[0]
        if key not in done:
            done.add(key)
            ps, pd = ctx.planes_alloc(S, S, 16, c["bd"], 1), ctx.planes_alloc(S, S, 16, c["bd"], 1)
            ctx.planes_upload(ps, 0, src); ctx.planes_upload(pd, 0, dst)
            if c["kind"] == "8x8":
                d_s, d_m = ctx.malloc(2 * 10 * 10), ctx.malloc(4 * 5 * 5)
                ctx.memset(d_s, 0x7f, 200); ctx.memset(d_m, 0x7f, 100)
                ctx.vbp_8x8_stats_plane(ps, 0, pd, 0, c["pw"], c["ph"], d_s, 10, d_m, 5)
                got = (ctx.from_device(d_s, (10, 10), np.int16), ctx.from_device(d_m, (5, 5), np.int32))
                ctx.free(d_s); ctx.free(d_m)
            else:
                d_s = ctx.malloc(2 * 20 * 20)
                ctx.memset(d_s, 0x7f, 800)
                ctx.vbp_4x4_avg_plane(ps, 0, c["pw"], c["ph"], c["border_offset"], d_s, 20)
                got = (ctx.from_device(d_s, (20, 20), np.int16),)
                ctx.free(d_s)
            ctx.planes_free(ps); ctx.planes_free(pd)
            cache = got
        if c["kind"] == "8x8":
            for k in range(4):
                x8, y8 = c["x16"] + 8 * (k & 1), c["y16"] + 8 * (k >> 1)
                if x8 < c["pw"] and y8 < c["ph"]:
                    assert int(cache[0][y8 // 8, x8 // 8]) == c["sum"][k] and c["sse"][k] == c["sum"][k] ** 2, (c, k)
                    n8 += 1
                else:
                    assert c["sum"][k] == 0 and c["sse"][k] == 0
                    if c["x16"] < c["pw"] and c["y16"] < c["ph"]:   # a leaf of a launched block beyond the visible part: written, 0
                        assert int(cache[0][y8 // 8, x8 // 8]) == 0, (c, k)
            if c["x16"] < c["pw"] and c["y16"] < c["ph"]:
                assert int(cache[1][c["y16"] // 16, c["x16"] // 16]) == c["minmax"], c
        else:
            for k in range(4):
                x4, y4 = c["x8"] + 4 * (k & 1), c["y8"] + 4 * (k >> 1)
                if x4 < c["pw"] and y4 < c["ph"]:
                    assert int(cache[0][y4 // 4, x4 // 4]) == c["sum"][k], (c, k)
                    n4 += 1
                else:
                    assert c["sum"][k] == 0
    assert n8 >= 60 and n4 >= 40


@pytest.mark.parametrize("bd,W,H,vw,vh", [(8, 256, 192, 256, 192), (8, 200, 136, 196, 132), (10, 208, 144, 202, 139), (12, 96, 80, 90, 72)])
def test_whole_planes_equal_the_oracle(hip, oracle, ctx, bd, W, H, vw, vh):
    rng = np.random.default_rng(bd * 31 + W)
    mx = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    src = rng.integers(0, mx + 1, (H, W)).astype(dt)
    dst = np.clip(src.astype(np.int64) + rng.integers(-(mx >> 2), (mx >> 2) + 1, (H, W)), 0, mx).astype(dt)
    dst[40:72, 40:72] = src[40:72, 40:72]
    B = 16
    ps, pd = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pd, 0, dst)
    sb, db = np.pad(src, B, mode="edge"), np.pad(dst, B, mode="edge")
    n8x, n8y, n16x, n16y = (vw + 7) // 8, (vh + 7) // 8, (vw + 15) // 16, (vh + 15) // 16
    # (all four leaves of every 16 x 16 block are written: 2 n16y rows of >= 2 n16x entries, no pre-zeroing needed)
    d_s, d_m = ctx.malloc(2 * (2 * n16x + 3) * 2 * n16y), ctx.malloc(4 * (n16x + 1) * n16y)
    ctx.memset(d_s, 0x7f, 2 * (2 * n16x + 3) * 2 * n16y)
    ctx.vbp_8x8_stats_plane(ps, 0, pd, 0, vw, vh, d_s, 2 * n16x + 3, d_m, n16x + 1)
    s8, m16 = ctx.from_device(d_s, (2 * n16y, 2 * n16x + 3), np.int16), ctx.from_device(d_m, (n16y, n16x + 1), np.int32)
    assert np.all(s8[n8y:, :2 * n16x] == 0) and np.all(s8[:, n8x:2 * n16x] == 0)   # the leaves beyond the visible part
    assert np.all(s8[:, 2 * n16x:] == 0x7f7f)                                          # nothing past the 2 n16x columns
    for y16 in range(0, vh, 16):
        for x16 in range(0, vw, 16):
            s, q, mm = oracle_8x8(sb, db, 0, 0, int(bd > 8), vw - x16, vh - y16, x0=B + x16, y0=B + y16)
            for k in range(4):
                x8, y8 = x16 + 8 * (k & 1), y16 + 8 * (k >> 1)
                if x8 < vw and y8 < vh:
                    assert int(s8[y8 // 8, x8 // 8]) == s[k], (x8, y8)
            assert int(m16[y16 // 16, x16 // 16]) == mm, (x16, y16)
    n4x, n4y = (vw + 3) // 4, (vh + 3) // 4
    d_4 = ctx.malloc(2 * n4x * n4y)
    for bo in (0, 4):
        ctx.vbp_4x4_avg_plane(ps, 0, vw, vh, bo, d_4, n4x)
        s4 = ctx.from_device(d_4, (n4y, n4x), np.int16)
        for y8 in range(0, vh, 8):
            for x8 in range(0, vw, 8):
                s, q = oracle_4x4(sb, 0, 0, int(bd > 8), vw - x8, vh - y8, bo, x0=B + x8, y0=B + y8)
                for k in range(4):
                    x4, y4 = x8 + 4 * (k & 1), y8 + 4 * (k >> 1)
                    if x4 < vw and y4 < vh:
                        assert int(s4[y4 // 4, x4 // 4]) == s[k], (x4, y4, bo)
    for d in (d_s, d_m, d_4):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pd)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p8, p10, pn = ctx.planes_alloc(64, 64, 16, 8, 1), ctx.planes_alloc(64, 64, 16, 10, 1), ctx.planes_alloc(64, 64, 4, 8, 1)
    d = ctx.malloc(4096)
    with pytest.raises(capi.AomHipError):
        ctx.vbp_8x8_stats_plane(p8, 0, p10, 0, 64, 64, d, 8)
    with pytest.raises(capi.AomHipError):
        ctx.vbp_8x8_stats_plane(p8, 0, p8, 0, 72, 64, d, 9)
    with pytest.raises(capi.AomHipError):
        ctx.vbp_8x8_stats_plane(p8, 0, p8, 0, 64, 64, d, 7)
    with pytest.raises(capi.AomHipError):
        ctx.vbp_8x8_stats_plane(p8, 0, p8, 0, 40, 40, d, 5)   # 2 * ceil(40 / 16) = 6 entries per row
    with pytest.raises(capi.AomHipError):
        ctx.vbp_8x8_stats_plane(p8, 0, pn, 0, 64, 64, d, 8)
    with pytest.raises(capi.AomHipError):
        ctx.vbp_4x4_avg_plane(p8, 0, 64, 64, -1, d, 16)
    ctx.free(d)
    for p_ in (p8, p10, pn):
        ctx.planes_free(p_)
