"""Mechanical boundary check: every exact-signature entry point of include/aomhip.h whose name exists in the reference's
rtcd definition files must have the reference's return type and parameter types, compared token by token after the preprocessor has
expanded the header's stamping macros.  The prototypes are read from the `add_proto` lines of aom_dsp/aom_dsp_rtcd_defs.pl and
av1/common/av1_rtcd_defs.pl where they lie under /root/reference (skipped where the reference is absent, e.g. on the GPU box)."""
import os
import re
import subprocess

import pytest

from conftest import BLOCK_SIZES, REFERENCE, ROOT, have_reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="needs /root/reference")

PL_FILES = ["aom_dsp/aom_dsp_rtcd_defs.pl", "av1/common/av1_rtcd_defs.pl"]
TYPE_WORDS = {"int", "char", "short", "long", "unsigned", "signed", "void", "float", "double", "const", "struct"}
# the reference's typedefs -> the C types the header spells out
ALIASES = {"tran_low_t": "int32_t", "uint32_t": "unsigned int", "unsigned": "unsigned int", "TX_TYPE": "int", "TxfmParam": "void",
           "DIST_WTD_COMP_PARAMS": "void", "CONV_BUF_TYPE": "uint16_t"}


def _tokens(decl):
    return re.findall(r"[A-Za-z_]\w*|\*|\[[^\]]*\]", decl)


def _param_type(p):
    """'const uint8_t * const ref_ptr[4]' -> 'const uint8_t * const *' (name dropped, array decays to pointer)."""
    tk = _tokens(p.strip())
    if not tk:
        return ""
    arr = 0
    while tk and tk[-1].startswith("["):
        tk.pop(); arr += 1
    has_type = any((t in TYPE_WORDS and t != "const") or t.endswith("_t") or t[0].isupper() for t in tk[:-1])
    if tk and re.match(r"[A-Za-z_]\w*$", tk[-1]) and has_type and tk[-1] not in TYPE_WORDS and not tk[-1].endswith("_t"):
        tk.pop()  # the parameter's name
    tk += ["*"] * arr
    out = []
    for t in tk:
        t = ALIASES.get(t, t)
        if t == "unsigned int" and out and out[-1] == "unsigned int":
            continue
        out.append(t)
    s = " ".join(out)
    s = s.replace("unsigned int int", "unsigned int")
    return s


def _signature(ret, args):
    ps = [] if args.strip() in ("", "void") else [_param_type(p) for p in args.split(",")]
    return " ".join(ALIASES.get(t, t) for t in _tokens(ret)).replace("unsigned int int", "unsigned int"), ps


def reference_protos():
    protos = {}
    for rel in PL_FILES:
        text = open(os.path.join(REFERENCE, rel)).read()
        for m in re.finditer(r'add_proto\s+qw/([^/]*)/\s*,\s*(?:"([^"]+)"\s*,\s*)?"([^"]*)"\s*;', text):
            head, name, args = m.group(1).split(), m.group(2), m.group(3)
            if name is None:
                name, head = head[-1], head[:-1]
            ret = " ".join(head)
            names = [name]
            if "${w}" in name or "${h}" in name:
                names = [name.replace("${w}", str(w)).replace("${h}", str(h)) for w, h in BLOCK_SIZES]
            for n in names:
                protos[n] = _signature(ret, args)  # (a name declared twice: the later add_proto wins, build/cmake/rtcd.pl add_proto)
    return protos


def header_protos():
    src = subprocess.run(["gcc", "-E", "-P", os.path.join(ROOT, "include", "aomhip.h")], check=True, capture_output=True, text=True).stdout
    out = {}
    for m in re.finditer(r"\b((?:const\s+)?(?:unsigned\s+int|unsigned|void|int|long|uint32_t|uint64_t|int64_t|char)\s*\*?)\s*(aomhip_\w+)\s*\(([^()]*)\)\s*;", src, re.S):
        out[m.group(2)] = _signature(m.group(1), m.group(3))
    return out


def test_exact_signature_entry_points_match_add_proto():
    ref, hdr = reference_protos(), header_protos()
    assert len(ref) > 1000 and len(hdr) > 200, (len(ref), len(hdr))
    checked, bad = [], []
    for name, sig in sorted(hdr.items()):
        stem = name[len("aomhip_"):]
        for cand in ("aom_" + stem, "av1_" + stem, stem):
            if cand in ref:
                checked.append(cand)
                if ref[cand] != sig:
                    bad.append((cand, "reference", ref[cand], "aomhip.h", sig))
                break
    assert not bad, bad
    fams = {"quantize": 12, "lpf": 40, "fwd_txfm2d": 19, "inv_txfm2d_add": 19, "cdef": 10, "subtract": 2, "sad": 2, "variance": 1}
    for key, n in fams.items():
        got = [c for c in checked if key in c]
        assert len(got) >= n, (key, len(got), n)
    assert len(checked) >= 105, len(checked)


def test_vtable_members_match_the_block_size_protos():
    """aomhip_variance_vtable's members carry the signatures of the per-block-size protos they replace (sdf <- aom_sadWxH, sdx4df <-
    aom_sadWxHx4d, vf <- aom_varianceWxH, svf <- aom_sub_pixel_varianceWxH, sdaf, svaf, sdsf, sdsx4df)."""
    ref = reference_protos()
    src = subprocess.run(["gcc", "-E", "-P", os.path.join(ROOT, "include", "aomhip.h")], check=True, capture_output=True, text=True).stdout
    body = re.search(r"typedef struct aomhip_variance_vtable \{(.*?)\} aomhip_variance_vtable;", src, re.S).group(1)
    members = {m.group(2): _signature(m.group(1), m.group(3)) for m in re.finditer(r"((?:unsigned int|void))\s*\(\*(\w+)\)\(([^()]*)\)\s*;", body)}
    pairs = {"sdf": "aom_sad16x16", "sdsf": "aom_sad_skip_16x16", "sdaf": "aom_sad16x16_avg", "vf": "aom_variance16x16",
             "svf": "aom_sub_pixel_variance16x16", "svaf": "aom_sub_pixel_avg_variance16x16", "sdx4df": "aom_sad16x16x4d",
             "sdx3df": "aom_sad16x16x3d", "sdsx4df": "aom_sad_skip_16x16x4d", "jsdaf": "aom_dist_wtd_sad16x16_avg",
             "jsvaf": "aom_dist_wtd_sub_pixel_avg_variance16x16"}
    for member, proto in pairs.items():
        assert member in members and proto in ref
        assert members[member] == ref[proto], (member, members[member], proto, ref[proto])
