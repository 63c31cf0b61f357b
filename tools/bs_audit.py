"""Block sizes against each other: the search kernels of one 4K frame pair cut into 8x8 .. 64x64 blocks, at 10 and 8 bits (timing only).
The work per PIXEL of a search is roughly constant over block sizes (fewer, larger blocks), so ms per frame should fall or stay as blocks grow; a
size that is much slower per frame than its neighbours points at an instantiation that spills, lost its window or serialised its reads.
  python tools/bs_audit.py [--sizes 8,16,32,64] [--reps 8]     ->  one JSON line per (depth, size)
Legs: DIAMOND step_param 4 (lean diamond kernel), the bilinear sub-pel tree (lean sub-pel kernel), NSTEP step_param 3 with cost list and second-best MV
(general search kernel), NSTEP + the good-quality mesh on every block, the 8-tap sub-pel tree (general sub-pel kernel)."""
import argparse, json, os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="8,16,32,64")
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    import importlib
    pkg = importlib.import_module("aom-av1-psy_amd")
    from benchlib import search, common
    capi = pkg.capi
    ctx = capi.Context(0)
    for bd in (10, 8):
        for bs in (int(v) for v in a.sizes.split(",")):
            class P(search.SearchPipeline):
                BS, BD = bs, bd
            wl = P(pkg, ctx, None, 0, 1, frames=2)
            n = wl.n
            d_cl, d_sec = ctx.malloc(n * 20), ctx.malloc(n * 4)
            q = capi.SearchParams.make("NSTEP", 3, capi.MV_COST_L1_HDRES)
            qm = capi.SearchParams.make("NSTEP", 3, capi.MV_COST_L1_HDRES, run_mesh=1, mesh=[(64, 8), (28, 4), (15, 1), (7, 1)])
            sp8 = capi.SubpelParams(2, capi.MV_COST_NONE, 0, 2, 1, 0, 3)
            for f in range(wl.F):
                wl.d_sub_blocks(f)
            legs = {
                "diamond": lambda f: ctx.fullpel_diamond_batch(wl.src, wl.ref, f, bs, bs, 0, 4, capi.MV_COST_L1_HDRES, wl.d_blocks, n, wl.d_mv, wl.d_cost),
                "subpel_bilinear": lambda f: ctx.subpel_bilinear_batch(wl.src, wl.ref, f, bs, bs, capi.MV_COST_L1_HDRES, 2, 1, 0, wl.d_sub_blocks(f), n, wl.d_smv, wl.d_err, wl.d_dist, wl.d_sse),
                "nstep_general": lambda f: ctx.full_pixel_search_batch(wl.src, wl.ref, f, bs, bs, q, wl.d_blocks, n, wl.d_mv, wl.d_cost, d_cl, d_sec),
                "nstep_mesh": lambda f: ctx.full_pixel_search_batch(wl.src, wl.ref, f, bs, bs, qm, wl.d_blocks, n, wl.d_mv, wl.d_cost),
                "subpel_tree_8tap": lambda f: ctx.subpel_tree_batch(wl.src, wl.ref, f, bs, bs, sp8, wl.d_sub_blocks(f), n, wl.d_smv, wl.d_err, wl.d_dist, wl.d_sse),
            }
            out = {"bd": bd, "bs": bs, "blocks": n}
            for name, fn in legs.items():
                k = [0]
                def once():
                    fn(k[0] % wl.F); k[0] += 1
                try:
                    common.ramp(ctx, once, 0.1)
                    out[name + "_ms"] = round(common.kernel_avg_ms(ctx, once, a.reps), 4)
                except Exception as e:
                    out[name + "_ms"] = "%s: %s" % (type(e).__name__, e)
            print(json.dumps(out), flush=True)
            ctx.free(d_cl); ctx.free(d_sec); wl.free()


if __name__ == "__main__":
    main()
