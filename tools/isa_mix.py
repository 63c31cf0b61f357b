#!/usr/bin/env python3
"""Static VALU opcode mix of every kernel of the library, by issue class -> profiles/r05_isa_mix.json.

    bash tools/list_scratch_kernels.sh /tmp/aomhip_asm   # device assembly of every csrc/*.hip (no GPU needed)
    python3 tools/isa_mix.py /tmp/aomhip_asm

Issue classes are the ones aomhip_valu_issue_probe measures on MI355X (profiles/r05_valu_issue.md):
  fast  : one wave64 instruction per ~2 clocks per SIMD (v_add/sub_u32, v_mov_b32, v_and/or/xor_b32, v_lshrrev_b32, v_ashrrev_i32,
          v_add/sub_u16, v_add/sub/mul/fma_f32, VOP2 and VOP3 encodings alike, and the 16/32-bit v_cmp_*)
  slow  : one per ~4 clocks (every other integer / packed / dot / SAD / DPP / 64-bit / fp64 / conversion instruction that was measured, and
          everything not measured)
  trans : one per ~8 clocks (v_exp/log/rcp/rsq/sqrt/sin/cos_f32 and _f16)
  trans64: one per ~16 clocks (v_rcp/rsq/sqrt_f64)
The mix is STATIC (instructions in the kernel's text, every loop body counted once): bench.py uses the class SHARES as weights for the dynamic
SQ_INSTS_VALU count of the PMC passes; a kernel whose hot loop has a different mix than its prologue is mis-weighted by that much."""
import glob, json, os, re, subprocess, sys
from collections import Counter, defaultdict

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u16",
        "v_sub_u16", "v_subrev_u16", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_not_b32"}
TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f32|f16)$")
TRANS64 = re.compile(r"^v_(rcp|rsq|sqrt)_f64$")


def classify(m, enc):
    if TRANS.match(m):
        return "trans"
    if TRANS64.match(m):
        return "trans64"
    if enc in ("sdwa", "dpp", "e64_dpp"):     # (DPP forms were measured at the 4-clock rate)
        return "slow"
    if m in FAST or re.match(r"^v_cmpx?_\w+_(i|u|f)(16|32)$", m):   # v_cmp + v_cndmask measured as a pair: 2 + 4 clocks
        return "fast"
    return "slow"


def main(d):
    out = {}
    names = []
    for f in sorted(glob.glob(os.path.join(d, "*.s"))):
        cur = None
        for ln in open(f):
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                cur = m.group(1)
                out[cur] = {"file": os.path.basename(f), "ops": Counter()}
                names.append(cur)
                continue
            if ln.startswith(".Lfunc_end"):
                cur = None
                continue
            if cur is None:
                continue
            m = re.match(r"^\s+(v_\w+)", ln)
            if m:
                op = m.group(1)
                enc = ""
                for suf in ("_e32", "_e64", "_sdwa", "_dpp", "_e64_dpp"):
                    if op.endswith(suf):
                        op, enc = op[:-len(suf)], suf[1:]
                        break
                if op in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32") or op.startswith("v_accvgpr") or op.startswith("v_mfma"):
                    cls = "slow"
                else:
                    cls = classify(op, enc)
                out[cur]["ops"][(op, enc, cls)] += 1
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for mangled, pretty in zip(names, dem):
        e = out[mangled]
        if not e["ops"] or "_kernel" not in pretty:
            continue
        tot = sum(e["ops"].values())
        by = defaultdict(int)
        for (op, enc, cls), n in e["ops"].items():
            by[cls] += n
        short = re.sub(r"^void ", "", pretty.split("(")[0]).replace("aomhip::", "").replace("(anonymous namespace)::", "")
        top = Counter()
        for (op, enc, cls), n in e["ops"].items():
            top[op] += n
        res[short] = {"file": e["file"], "valu_static": tot, "share": {k: by[k] / tot for k in ("fast", "slow", "trans", "trans64") if by[k]},
                      "top": dict(top.most_common(8))}
    json.dump(res, open(os.path.join(d, "isa_mix_all.json"), "w"), indent=0, sort_keys=True)     # every kernel: scratch copy beside the assembly
    # committed: the kernels some PMC pass under profiles/ has instruction counts for (bench.py weights those counts with these shares)
    want = set()
    for f in glob.glob("profiles/r0*_pmc.json"):
        try:
            want.update(k for k in json.load(open(f)) if "kernel" in k)
        except Exception:
            pass
    keep = {k: v for k, v in res.items() if any(k.startswith(w.rstrip(">").rstrip()) or w.startswith(k) for w in want)}
    json.dump({"_doc": __doc__.split("\n\n")[2], "kernels": keep}, open("profiles/r05_isa_mix.json", "w"), indent=0, sort_keys=True)
    print(len(keep), "kernels kept")
    for k in sorted(res):
        if any(s in k for s in ("diamond_kernel<unsigned short, 16, 16, 2, true", "subpel_bilinear_kernel<unsigned short, 16, 16", "cdef_luma_kernel<unsigned short",
                                "xform_quant_staged_kernel<16, 16, true", "sad_strip_kernel<unsigned char, 16, 16", "tf_apply", "inter_pred_kernel<unsigned short, 16, 16")):
            print(k[:90], res[k]["valu_static"], {a: round(b, 3) for a, b in res[k]["share"].items()})
    print(len(res), "kernels")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/aomhip_asm")
