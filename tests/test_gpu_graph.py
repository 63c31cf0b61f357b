"""aomhip_graph_*: a recorded sequence of batched calls replays with the same results, picks up new DATA at frozen addresses, and a capture
that would allocate is refused with an error instead of a crash."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_replayed_chain_equals_the_direct_calls(hip, oracle, ctx):
    capi = hip.capi
    W, H, B, bs, bd = 256, 128, 64, 16, 10
    src, ref = hip.synth.shifted_smooth_pair(W, H, 2, bd, shift=(3, -2), frac8=(2, 6))
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = B - 8
    blocks["col_min"] = np.maximum(-(blocks["bx"] + ext), -1023); blocks["col_max"] = np.minimum(W - blocks["bx"] - bs + ext, 1023)
    blocks["row_min"] = np.maximum(-(blocks["by"] + ext), -1023); blocks["row_max"] = np.minimum(H - blocks["by"] - bs + ext, 1023)
    d_b = ctx.to_device(blocks)
    d_mv, d_cost = ctx.malloc(n * 4), ctx.malloc(n * 4)
    d_sb = ctx.malloc(n * blocks.itemsize)
    d_smv, d_err, d_dist, d_sse = (ctx.malloc(n * 4) for _ in range(4))

    def seq():   # full-pel search, then an element-wise fill behind it: two stream-ordered operations whose order the graph must keep
        ctx.fullpel_diamond_batch(ps, pr, 0, bs, bs, 0, 4, capi.MV_COST_L1_HDRES, d_b, n, d_mv, d_cost)
        ctx.memset(d_dist, 0x5A, n * 4)

    seq(); ctx.sync()
    want_mv, want_cost = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_cost, (n,), np.int32)
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    omv, ocost = oracle.fullpel_diamond_batch(sb, rb, B, bs, bs, blocks, 0, 4, 3, bd)
    assert np.array_equal(want_mv, omv) and np.array_equal(want_cost, ocost)
    g = ctx.capture(seq)
    ctx.memset(d_mv, 0, n * 4); ctx.memset(d_cost, 0, n * 4); ctx.memset(d_dist, 0, n * 4)
    ctx.graph_launch(g); ctx.sync()
    assert np.array_equal(ctx.from_device(d_mv, (n, 2), np.int16), want_mv) and np.array_equal(ctx.from_device(d_cost, (n,), np.int32), want_cost)
    assert (ctx.from_device(d_dist, (n,), np.uint32) == 0x5A5A5A5A).all()
    # new data at the same addresses: the replay searches the new frame
    src2, ref2 = hip.synth.shifted_smooth_pair(W, H, 9, bd, shift=(-4, 5), frac8=(0, 0))
    ctx.planes_upload(ps, 0, src2); ctx.planes_upload(pr, 0, ref2)
    ctx.graph_launch(g); ctx.sync()
    sb2, rb2 = oracle.extend_plane(src2, B, ps.stride), oracle.extend_plane(ref2, B, pr.stride)
    omv2, ocost2 = oracle.fullpel_diamond_batch(sb2, rb2, B, bs, bs, blocks, 0, 4, 3, bd)
    assert np.array_equal(ctx.from_device(d_mv, (n, 2), np.int16), omv2) and np.array_equal(ctx.from_device(d_cost, (n,), np.int32), ocost2)
    assert not np.array_equal(omv2, omv)
    ctx.graph_destroy(g)
    for d in (d_b, d_mv, d_cost, d_sb, d_smv, d_err, d_dist, d_sse):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_a_capture_that_synchronises_is_refused(hip):
    """A blocking call inside a capture is an error, not a crash; HIP leaves that stream in its 'capture invalidated' state, so the documented
    way on is a new context (include/aomhip.h)."""
    capi = hip.capi
    bad_ctx = capi.Context(0)
    d = bad_ctx.malloc(64)

    def bad():
        bad_ctx.memset(d, 1, 64)
        bad_ctx.from_device(d, (16,), np.int32)   # a blocking copy: not capturable

    with pytest.raises(capi.AomHipError):
        bad_ctx.capture(bad)
    ctx2 = capi.Context(0)                        # other contexts (other streams) are unaffected
    e = ctx2.malloc(64)
    ctx2.memset(e, 2, 64); ctx2.sync()
    assert (ctx2.from_device(e, (16,), np.int32) == 0x02020202).all()
    ctx2.free(e)
    ctx2.close()


def test_a_graph_is_refused_after_the_contexts_work_memory_moved(hip):
    """The composite entry points keep intermediates in the context's work memory, which is freed and reallocated when a later call needs
    more: a graph captured before that froze the old address.  aomhip_graph_launch must refuse it (include/aomhip.h) instead of replaying
    into freed memory; a graph captured afterwards works."""
    capi = hip.capi
    ctx = capi.Context(0)
    W, H, B, bs, bd = 256, 128, 64, 16, 8
    src, ref = hip.synth.shifted_smooth_pair(W, H, 4, bd, shift=(2, -3), frac8=(0, 0))
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = B - 8
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bs + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bs + ext
    big = np.tile(blocks, 16)                      # the same blocks 16 times over: 16 x the work memory
    d_small, d_big = ctx.to_device(blocks), ctx.to_device(big)
    outs = [ctx.malloc(big.size * 4) for _ in range(5)]
    full = capi.SearchParams.make("NSTEP", 2, capi.MV_COST_L1_HDRES)
    sub = capi.SubpelParams(capi.SUBPEL_TREES["pruned"], capi.MV_COST_NONE, 64, 2, 1, 0, 0)
    me = lambda d, m: ctx.motion_estimation_batch(ps, pr, 0, bs, bs, full, sub, 0, d, m, *outs)
    me(d_small, n); ctx.sync()
    want = ctx.from_device(outs[0], (n, 2), np.int16).copy()
    g = ctx.capture(lambda: me(d_small, n))
    ctx.graph_launch(g); ctx.sync()                # valid: nothing moved yet
    assert np.array_equal(ctx.from_device(outs[0], (n, 2), np.int16), want)
    me(d_big, big.size); ctx.sync()                # grows the work memory: free + malloc
    with pytest.raises(capi.AomHipError, match="capture again"):
        ctx.graph_launch(g)
    g2 = ctx.capture(lambda: me(d_small, n))       # captured on the grown buffers
    ctx.memset(outs[0], 0, n * 4)
    ctx.graph_launch(g2); ctx.sync()
    assert np.array_equal(ctx.from_device(outs[0], (n, 2), np.int16), want)
    other = capi.Context(0)
    with pytest.raises(capi.AomHipError, match="another context"):
        other.graph_launch(g2)
    ctx.graph_destroy(g); ctx.graph_destroy(g2)


def test_first_pass_frame_call_replays_from_a_graph_with_and_without_its_side_stream(hip):
    """aomhip_first_pass_inter_frame forks its golden-frame leg onto the context's second stream and joins it again inside the call: a capture of
    ctx->stream must record the fork and the join (cross-stream events are capturable), and a capture taken before the second stream exists must
    not try to create it (stream creation is not capturable) -- that replay runs everything on one stream.  Both graphs and the direct call give
    the same five outputs."""
    capi = hip.capi
    ctx = capi.Context(0)   # a fresh context: no side stream yet
    W, H, B, bs, bd = 352, 288, 64, 16, 8
    rng = np.random.default_rng(5)
    src, last = hip.synth.shifted_smooth_pair(W, H, 11, bd, shift=(5, -7), frac8=(0, 0))
    _, gold = hip.synth.shifted_smooth_pair(W, H, 11, bd, shift=(-2, 3), frac8=(0, 0))
    noisy = lambda a, k: np.clip(a.astype(np.int32) + rng.integers(-k, k + 1, a.shape), 0, 255).astype(a.dtype)
    last, gold, lsrc = noisy(last, 3), noisy(gold, 5), noisy(last, 6)
    rings = [ctx.planes_alloc(W, H, B, bd, 1) for _ in range(4)]
    for ring, img in zip(rings, (src, last, gold, lsrc)):
        ctx.planes_upload(ring, 0, img)
    ps, pl, pg, pls = rings
    rows, cols = H // bs, W // bs
    n = rows * cols
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % cols) * bs, (np.arange(n) // cols) * bs
    ext = B - 8
    blocks["col_min"] = np.maximum(-(blocks["bx"] + ext), -1023); blocks["col_max"] = np.minimum(W - blocks["bx"] - bs + ext, 1023)
    blocks["row_min"] = np.maximum(-(blocks["by"] + ext), -1023); blocks["row_max"] = np.minimum(H - blocks["by"] - bs + ext, 1023)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    tj, t0, t1 = np.array([200, 650, 640, 1050], np.int32), (150 + bits * 310).astype(np.int32), (170 + bits * 290 + (v & 7) * 3).astype(np.int32)
    q = capi.SearchParams.make("NSTEP_FPF", 2, 0, sad_per_bit=24, error_per_bit=70)
    fp = capi.FirstPassParams(rows, cols, 0, 0)
    intra = rng.integers(2000, 60000, n).astype(np.int32)
    d_b, d_i, d_j, d_c0, d_c1 = ctx.to_device(blocks), ctx.to_device(intra), ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    outs = [ctx.malloc(n * 4) for _ in range(5)]

    def call():
        ctx.first_pass_inter_frame(ps, 0, pl, 0, pg, 0, pls, 0, bs, bs, q, fp, d_b, d_i, outs[0], outs[2], outs[1], outs[3], outs[4], d_j, d_c0 + mv_max * 4,
                                   d_c1 + mv_max * 4)

    def read():
        ctx.sync()
        return [ctx.from_device(o, (n,), np.int32).copy() for o in outs]

    def clear():
        for o in outs:
            ctx.memset(o, 0xEE, n * 4)

    # the work memory must exist before a capture (a capture that would allocate is refused): size it with the side stream still absent
    import os
    os.environ["AOMHIP_FP_SERIAL"] = "1"
    try:
        call()
    finally:
        os.environ["AOMHIP_FP_SERIAL"] = "0"
    want = read()
    g_serial = ctx.capture(call)          # no side stream yet: captured on one stream
    clear(); ctx.graph_launch(g_serial)
    got = read()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    clear(); call()                       # outside a capture: creates the side stream, forks and joins
    got = read()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    g_forked = ctx.capture(call)          # now the fork / join is part of the graph
    clear(); ctx.graph_launch(g_forked)
    got = read()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    clear(); ctx.graph_launch(g_serial)   # and the first graph still replays
    got = read()
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert (want[0] != 0).any()
    ctx.graph_destroy(g_serial); ctx.graph_destroy(g_forked)
    for d in [d_b, d_i, d_j, d_c0, d_c1] + outs:
        ctx.free(d)
    for r in rings:
        ctx.planes_free(r)
    ctx.close()


def test_tf_frames_call_gives_the_same_outputs_on_one_stream_on_two_and_from_graphs(hip):
    """aomhip_tf_motion_search_frames runs a frame's 16x16 searches on the context's second stream beside the next frame's 32x32 search
    (csrc/tf_search.hip): AOMHIP_TF_SERIAL=1, the forked form, a capture taken before the second stream exists (one stream) and a capture with
    the fork / joins recorded all write the same MVs, errors and ref_mvs."""
    import os
    capi = hip.capi
    ctx = capi.Context(0)   # a fresh context: no side stream yet
    W, H, B, bd, F = 352, 288, 64, 10, 5
    rng = np.random.default_rng(9)
    base, _ = hip.synth.shifted_smooth_pair(W + 64, H + 64, 17, bd)
    planes = ctx.planes_alloc(W, H, B, bd, F)
    for f in range(F):
        img = base[32 + f:32 + f + H, 32 - 2 * f:32 - 2 * f + W].astype(np.int32) + rng.integers(-3, 4, (H, W))
        ctx.planes_upload(planes, f, np.clip(img, 0, (1 << bd) - 1).astype(np.uint16))
    blocks = capi.tf_block_list(W, H, B)
    n = len(blocks)
    tp = capi.TfParams.default(W, H, bd, 30, 1, [(64, 8), (28, 4), (15, 1), (7, 1)])
    d_b = ctx.to_device(blocks)
    outs = [ctx.malloc(F * n * 16), ctx.malloc(F * n * 16), ctx.malloc(n * 4)]
    sizes = [F * n * 16, F * n * 16, n * 4]

    def call():
        ctx.tf_motion_search_frames(planes, F // 2, tp, d_b, n, outs[0], outs[1], outs[2])

    def read():
        ctx.sync()
        return [ctx.from_device(o, (s // 2,), np.int16).copy() for o, s in zip(outs, sizes)]

    def clear():
        for o, s in zip(outs, sizes):
            ctx.memset(o, 0xEE, s)

    os.environ["AOMHIP_TF_SERIAL"] = "1"
    try:
        call()                            # (sizes the work memory; one stream)
        want = read()
        g_serial = ctx.capture(call)      # no side stream yet
    finally:
        os.environ["AOMHIP_TF_SERIAL"] = "0"
    clear(); ctx.graph_launch(g_serial)
    assert all(np.array_equal(a, b) for a, b in zip(read(), want))
    clear(); call()                       # outside a capture: creates the side stream, forks per frame, joins once
    assert all(np.array_equal(a, b) for a, b in zip(read(), want))
    g_forked = ctx.capture(call)
    clear(); ctx.graph_launch(g_forked)
    assert all(np.array_equal(a, b) for a, b in zip(read(), want))
    clear(); ctx.graph_launch(g_serial)
    assert all(np.array_equal(a, b) for a, b in zip(read(), want))
    assert (want[0] != 0).any()
    ctx.graph_destroy(g_serial); ctx.graph_destroy(g_forked)
    for d in [d_b] + outs:
        ctx.free(d)
    ctx.planes_free(planes)
    ctx.close()
