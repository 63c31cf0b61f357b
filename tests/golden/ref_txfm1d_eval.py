"""Mechanical evaluator for the reference's straight-line 1-D transform sources.

TEST INFRASTRUCTURE (golden-vector generator) -- runs only in the build container,
where /root/reference exists.  It does not copy reference source: it *reads* the
reference's `av1/encoder/av1_fwd_txfm1d.c` / `av1/common/av1_inv_txfm1d.c`, parses
each butterfly function's stage statements (`bf1[i] = bf0[j] + bf0[k];`,
`bf1[i] = half_btf(w0, a, w1, b, cos_bit);` ...) into a per-stage op list and
evaluates that op list on seeded inputs with the exact integer semantics of
`half_btf` (av1/common/av1_txfm.h:80-102: 32-bit wrapping products, 64-bit sum,
rounding shift).  The outputs are committed as fixtures (tests/golden/txfm1d_*.npz)
that pin both the C oracle (oracle/) and the HIP kernels.

The reference cannot be *compiled* here under the project rules (every source
includes the cmake-generated config/aom_config.h), which is why the 1-D networks --
the part of the path where a transcription slip is most likely -- are pinned by
evaluating the reference's own statements instead.
"""
import re
import numpy as np

REF = "/root/reference"

_ASSIGN = re.compile(r"^bf1\[(\d+)\]\s*=\s*(.+)$")
_TERM = re.compile(r"^(-?)\s*(bf0|input)\[(\d+)\]$")
_W = re.compile(r"^(-?)cospi\[(\d+)\]$")


def _parse_term(t):
    m = _TERM.match(t.strip())
    if not m:
        raise ValueError("term? %r" % t)
    return (-1 if m.group(1) else 1, int(m.group(3)))


def _parse_expr(e):
    e = e.strip()
    if e.startswith("half_btf("):
        args = [a.strip() for a in e[len("half_btf("):-1].split(",")]
        assert len(args) == 5 and args[4] == "cos_bit", e
        w0 = _W.match(args[0]); w1 = _W.match(args[2])
        s0, i0 = _parse_term(args[1]); s1, i1 = _parse_term(args[3])
        assert s0 == 1 and s1 == 1
        return ("hb", (-1 if w0.group(1) else 1) * 1, int(w0.group(2)), i0,
                (-1 if w1.group(1) else 1) * 1, int(w1.group(2)), i1)
    if e.startswith("clamp_value("):
        inner, rng = e[len("clamp_value("):-1].rsplit(",", 1)
        assert rng.strip() == "stage_range[stage]", e
        op = _parse_expr(inner)
        assert op[0] == "add"
        return ("addc",) + op[1:]
    # a + b | a - b | -a + b | a | -a
    m = re.match(r"^(-?\s*\w+\[\d+\])\s*([+-])\s*(\w+\[\d+\])$", e)
    if m:
        sa, ia = _parse_term(m.group(1))
        sb, ib = _parse_term(m.group(3))
        if m.group(2) == "-":
            sb = -sb
        return ("add", sa, ia, sb, ib)
    s, i = _parse_term(e)
    return ("copy", s, i)


def parse_functions(path):
    """-> {name: [stage, ...]}, stage = list of (dst, op) covering every index."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"void\s+(av1_[fi](?:dct|adst)\d+)\s*\([^)]*\)\s*\{", src):
        name = m.group(1)
        # body up to the matching closing brace at column 0
        end = src.index("\n}\n", m.end())
        body = src[m.end():end]
        stmts = [re.sub(r"\s+", " ", s).strip() for s in body.split(";")]
        if not any(s.startswith("bf1[") for s in stmts):
            continue  # adst4 style: handled by hand in the oracle
        stages, cur = [], None
        for s in stmts:
            # a stage starts where the write pointer is (re)bound; `stage++` alone is not
            # reliable (av1_iadst8/16's last stage has no `stage++`).
            if re.match(r"^bf1 = (output|step)$", s):
                cur = {}
                stages.append(cur)
                continue
            a = _ASSIGN.match(s)
            if a:
                cur[int(a.group(1))] = _parse_expr(a.group(2))
        size = int(re.search(r"const int32_t size = (\d+)", body).group(1))
        for st in stages:
            assert sorted(st) == list(range(size)), (name, sorted(st))
        out[name] = (size, [[st[i] for i in range(size)] for st in stages])
    return out


def parse_int_table(path, name):
    """Parse `name[..][..] = { ... };` numeric initialiser -> flat list of ints."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    m = re.search(re.escape(name) + r"\s*(?:\[[^\]]*\])+\s*\)?\s*=\s*\{", src)
    if not m:
        raise KeyError(name)
    depth, i = 1, m.end()
    while depth:
        c = src[i]
        depth += (c == "{") - (c == "}")
        i += 1
    body = src[m.end():i - 1]
    return [int(x) for x in re.findall(r"-?\d+", body)]


def _wrap32(x):
    return ((x + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)


def half_btf(w0, a, w1, b, bit):
    """av1/common/av1_txfm.h:80-102 on int64 numpy arrays."""
    p0 = _wrap32(np.int64(w0) * a)
    p1 = _wrap32(np.int64(w1) * b)
    return _wrap32((p0 + p1 + (np.int64(1) << (bit - 1))) >> bit)


def evaluate(fn, x, cos_bit, cospi, clamp_bit=0):
    """x: int64 array [..., size]; cospi: the 64-entry row for cos_bit.
    clamp_bit: the (uniform) stage_range value the inverse transforms clamp their
    add/sub stages to (av1/common/av1_inv_txfm1d.h:21-26); <=0 disables."""
    size, stages = fn
    v = x.astype(np.int64)
    for st in stages:
        nv = np.empty_like(v)
        for dst, op in enumerate(st):
            if op[0] == "copy":
                nv[..., dst] = op[1] * v[..., op[2]]
            elif op[0] == "add":
                nv[..., dst] = _wrap32(op[1] * v[..., op[2]] + op[3] * v[..., op[4]])
            elif op[0] == "addc":
                t = _wrap32(op[1] * v[..., op[2]] + op[3] * v[..., op[4]])
                if clamp_bit > 0:
                    t = np.clip(t, -(1 << (clamp_bit - 1)), (1 << (clamp_bit - 1)) - 1)
                nv[..., dst] = t
            else:
                _, s0, k0, i0, s1, k1, i1 = op
                nv[..., dst] = half_btf(s0 * cospi[k0], v[..., i0], s1 * cospi[k1], v[..., i1], cos_bit)
        v = nv
    return v


def load_reference_networks():
    f = parse_functions(REF + "/av1/encoder/av1_fwd_txfm1d.c")
    f.update(parse_functions(REF + "/av1/common/av1_inv_txfm1d.c"))
    cospi = np.array(parse_int_table(REF + "/av1/common/av1_txfm.c", "av1_cospi_arr_data"),
                     dtype=np.int64).reshape(7, 64)
    sinpi = np.array(parse_int_table(REF + "/av1/common/av1_txfm.c", "av1_sinpi_arr_data"),
                     dtype=np.int64).reshape(7, 5)
    return f, cospi, sinpi


if __name__ == "__main__":
    f, cospi, sinpi = load_reference_networks()
    for k, (size, st) in sorted(f.items()):
        print(k, size, len(st))
