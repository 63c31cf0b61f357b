"""A small C-subset interpreter used to EVALUATE THE REFERENCE'S OWN FUNCTIONS where they lie.

TEST INFRASTRUCTURE (golden-vector generator) -- runs only in the build container, where /root/reference
exists.  Nothing of the reference is copied: `CEval.load(path)` reads a reference source or header file,
runs a miniature preprocessor over it (object- and function-like macros, #if / #ifdef on caller-supplied
configuration values), and keeps every function definition, typedef and global table it finds.
`CEval.call(name, *args)` then interprets the function body with C's integer semantics:

  * fixed-width integer types (LP64: int 32, long 64), integer promotions and the usual arithmetic
    conversions on every binary operator, wrap-around on conversion / overflow, arithmetic right shift of
    negative values, division truncating towards zero;
  * pointers as (buffer, element offset) pairs, so negative indexing (`s[-4 * pitch]`), pointer
    arithmetic, `*p`, `&x`, multi-dimensional arrays and pointers-to-rows work as in C; reads of
    uninitialised locals or out-of-bounds elements raise instead of returning garbage;
  * statements: declarations with initialisers, if / for / while / do / switch / break / continue / return.

Not supported (not needed by the functions evaluated): structs / unions, floating point, goto, function
pointers, variadic calls.  The reference cannot be *compiled* here under the project rules (every source
includes cmake-generated config headers), which is why its integer kernels are pinned by interpreting
their statements instead.  Generators built on this module: tests/golden/gen_ref_eval_golden.py.
"""
import re

# ----------------------------------------------------------------------------------------------------- types


class T:
    __slots__ = ("bits", "signed", "mask", "name", "size")

    def __init__(self, bits, signed, name):
        self.bits, self.signed, self.name = bits, signed, name
        self.mask = (1 << bits) - 1
        self.size = bits // 8

    def __repr__(self):
        return self.name


I8, U8 = T(8, True, "int8_t"), T(8, False, "uint8_t")
I16, U16 = T(16, True, "int16_t"), T(16, False, "uint16_t")
I32, U32 = T(32, True, "int"), T(32, False, "unsigned int")
I64, U64 = T(64, True, "int64_t"), T(64, False, "uint64_t")
F64 = T(64, True, "double")      # the only floating type: used by the site builders ((int)(0.41 * radius) ...)
F64.mask = None
PTR = "ptr"      # type tag of pointer-valued expressions
VOID = "void"

BASE_TYPEDEFS = {
    "int8_t": I8, "uint8_t": U8, "int16_t": I16, "uint16_t": U16, "int32_t": I32, "uint32_t": U32,
    "int64_t": I64, "uint64_t": U64, "intptr_t": I64, "uintptr_t": U64, "size_t": U64, "ptrdiff_t": I64,
    "ssize_t": I64,
    # <stdbool.h>: the interpreted paths only ever store comparison results (already 0 / 1) in a bool
    "bool": U8,
}
TYPE_WORDS = {"void", "char", "short", "int", "long", "signed", "unsigned", "_Bool", "double", "float"}
QUALIFIERS = {"const", "static", "inline", "__inline", "__inline__", "volatile", "register", "extern", "restrict",
              "__restrict", "INLINE", "AOM_INLINE", "AOM_FORCE_INLINE"}


def wrap(v, t):
    if t is F64:
        return float(v)
    if v.__class__ is float:
        v = int(v)                 # C conversion: truncation towards zero
    v &= t.mask
    if t.signed and v >> (t.bits - 1):
        v -= 1 << t.bits
    return v


def promote(t):
    return I32 if t.bits < 32 else t      # (F64 has 64 bits: unchanged)


def common(a, b):
    if a is F64 or b is F64:
        return F64
    a, b = promote(a), promote(b)
    if a is b:
        return a
    if a.bits == b.bits:
        return a if not a.signed else b      # the unsigned one
    big, small = (a, b) if a.bits > b.bits else (b, a)
    return big                                # 64-bit type can represent every 32-bit value (LP64)


def copy_struct(sv):
    f = {}
    for k, p in sv.f.items():
        buf = [copy_struct(x) if x.__class__ is StructVal else x for x in p.buf]
        f[k] = Ptr(buf, 0, p.t, p.dims)
    return StructVal(sv.st, f)


class Ptr:
    """C pointer / decayed array: element `off` of `buf`; `dims` = shape of the pointed-to sub-array (() = scalar).
    `t` is the element type: an integer T, a ("ptr", ...) type or a StructType."""
    __slots__ = ("buf", "off", "t", "dims", "stride")

    def __init__(self, buf, off, t, dims=()):
        self.buf, self.off, self.t, self.dims = buf, off, t, dims
        s = 1
        for d in dims:
            s *= d
        self.stride = s

    def add(self, n):
        return Ptr(self.buf, self.off + n * self.stride, self.t, self.dims)

    def deref(self):
        if self.dims:
            return Ptr(self.buf, self.off, self.t, self.dims[1:]), PTR
        if not 0 <= self.off < len(self.buf):
            raise CError("out-of-bounds read at element %d of %d" % (self.off, len(self.buf)))
        v = self.buf[self.off]
        t = self.t
        if t.__class__ is T:
            if v is None:
                raise CError("read of uninitialised element %d" % self.off)
            if v < 0:
                if not t.signed:
                    v += t.mask + 1          # element written through a signed view of the same width
            elif t.signed and t.mask is not None and v > (t.mask >> 1):
                v -= t.mask + 1
            return v, t
        if t.__class__ is StructType:
            return v, t
        if v is UNINIT:
            raise CError("read of uninitialised pointer")
        return v, PTR

    def store(self, v, vt):
        """C assignment into this element: converts integers, copies structs; returns the stored (value, tag)."""
        if self.dims:
            raise CError("store to array")
        if not 0 <= self.off < len(self.buf):
            raise CError("out-of-bounds write at element %d of %d" % (self.off, len(self.buf)))
        t = self.t
        if t.__class__ is T:
            if vt.__class__ is not T:
                raise CError("non-integer stored to %s" % t)
            v = wrap(v, t)
            self.buf[self.off] = v
            return v, t
        if t.__class__ is StructType:
            if vt is not t:
                # the two arms of a union the evaluator models as ONE member (ALIASED_STRUCTS, set by the generator that needs it: int_mv's
                # as_mv / as_fullmv, both { int16_t row; int16_t col }): the value is stored re-tagged, field for field
                if vt.__class__ is StructType and frozenset((t.name, vt.name)) in ALIASED_STRUCTS and [f[0] for f in t.fields] == [f[0] for f in vt.fields]:
                    c = copy_struct(v)
                    c.st = t
                    self.buf[self.off] = c
                    return c, t
                raise CError("struct type mismatch: %s <- %s" % (t, vt))
            self.buf[self.off] = copy_struct(v)
            return v, t
        if vt is not PTR:
            if vt.__class__ is not T or v != 0:
                raise CError("non-pointer stored to pointer")
            v = None
        self.buf[self.off] = v
        return v, PTR


class CError(Exception):
    pass


class StructType:
    """A struct: ordered (field name, type) list; `fields is None` while only forward-declared."""
    __slots__ = ("name", "fields")

    def __init__(self, name):
        self.name, self.fields = name, None

    def __repr__(self):
        return "struct " + self.name


ALIASED_STRUCTS = set()   # frozenset({name_a, name_b}) of layout-identical struct types that stand for the arms of one union


class StructVal:
    """One struct object: field name -> storage Ptr (so &s.f, s.f[i] and nested structs work like variables)."""
    __slots__ = ("st", "f")

    def __init__(self, st, f):
        self.st, self.f = st, f


class FuncRef:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name


class _Uninit:
    def __repr__(self):
        return "<uninitialised pointer>"


UNINIT = _Uninit()


# ------------------------------------------------------------------------------------------------- tokenizer

_TOK = re.compile(r"""
    (?P<ws>\s+)
  | (?P<num>0[xX][0-9a-fA-F]+[uUlL]*|\d+\.\d*(?:[eE][-+]?\d+)?[fFlL]?|\d+[eE][-+]?\d+[fFlL]?|\d+[uUlL]*)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])+')
  | (?P<op>\#\#|<<=|>>=|\.\.\.|->|\+\+|--|<<|>>|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|%=|&=|\|=|\^=|[-+*/%&|^~!<>=?:;,.(){}\[\]\#])
""", re.X)


def tokenize(text):
    out = []
    pos = 0
    n = len(text)
    while pos < n:
        m = _TOK.match(text, pos)
        if not m:
            raise CError("cannot tokenize %r" % text[pos:pos + 30])
        pos = m.end()
        k = m.lastgroup
        if k != "ws":
            out.append((k, m.group()))
    return out


def strip_comments(text):
    def rep(m):
        s = m.group()
        if s.startswith("/"):
            return " " if s.startswith("//") else " " + "\n" * s.count("\n")
        return s
    return re.sub(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])+\'', rep, text, flags=re.S)


# ----------------------------------------------------------------------------------------------- preprocessor

class Macro:
    __slots__ = ("params", "body")

    def __init__(self, params, body):
        self.params, self.body = params, body


class Preprocessor:
    def __init__(self, defines):
        self.macros = {}
        for k, v in defines.items():
            self.macros[k] = Macro(None, tokenize(str(v)))

    def expand(self, toks, hidden=frozenset()):
        out = []
        i, n = 0, len(toks)
        while i < n:
            k, s = toks[i]
            m = self.macros.get(s) if k == "id" else None
            if m is None or s in hidden:
                out.append(toks[i]); i += 1
                continue
            if m.params is None:
                out.extend(self.expand(m.body, hidden | {s})); i += 1
                continue
            if i + 1 >= n or toks[i + 1][1] != "(":
                out.append(toks[i]); i += 1
                continue
            j, depth, args, cur = i + 2, 1, [], []
            while j < n:
                tj = toks[j]
                if tj[1] in "([{" and tj[0] == "op":
                    depth += 1
                elif tj[1] in ")]}" and tj[0] == "op":
                    depth -= 1
                    if depth == 0:
                        break
                if depth == 1 and tj[1] == "," and tj[0] == "op":
                    args.append(cur); cur = []
                else:
                    cur.append(tj)
                j += 1
            if j >= n:
                raise CError("unterminated macro call %s" % s)
            if cur or args:
                args.append(cur)
            if len(args) != len(m.params):
                if not (len(m.params) == 0 and not args):
                    raise CError("macro %s: %d args for %d params" % (s, len(args), len(m.params)))
            amap = dict(zip(m.params, args))
            body, rep, b = m.body, [], 0
            while b < len(body):
                tb = body[b]
                if b + 1 < len(body) and body[b + 1][1] == "##":     # token pasting (raw arguments)
                    left = amap[tb[1]] if tb[0] == "id" and tb[1] in amap else [tb]
                    txt = "".join(x[1] for x in left)
                    while b + 1 < len(body) and body[b + 1][1] == "##":
                        nt = body[b + 2]
                        right = amap[nt[1]] if nt[0] == "id" and nt[1] in amap else [nt]
                        txt += "".join(x[1] for x in right)
                        b += 2
                    rep.extend(tokenize(txt)); b += 1
                    continue
                if tb[0] == "id" and tb[1] in amap:
                    rep.extend(self.expand(amap[tb[1]], hidden))
                else:
                    rep.append(tb)
                b += 1
            out.extend(self.expand(rep, hidden | {s}))
            i = j + 1
        return out

    def cond(self, toks):
        res, i = [], 0
        while i < len(toks):
            if toks[i][1] == "defined":
                if toks[i + 1][1] == "(":
                    name = toks[i + 2][1]; i += 4
                else:
                    name = toks[i + 1][1]; i += 2
                res.append(("num", "1" if name in self.macros else "0"))
            else:
                res.append(toks[i]); i += 1
        res = [("num", "0") if k == "id" else (k, s) for k, s in self.expand(res)]
        p = Parser(res, {})
        v, _ = Interp.const_eval(p.expr())
        return bool(v)

    def run(self, text):
        text = strip_comments(text).replace("\\\n", " ")
        out, pending = [], []
        stack = []          # (parent_active, this_branch_taken_already, currently_active)
        active = True
        for line in text.split("\n"):
            s = line.strip()
            if not s.startswith("#"):
                if active and s:
                    pending.extend(tokenize(line))
                continue
            if pending:
                out.extend(self.expand(pending)); pending = []
            m = re.match(r"#\s*(\w+)\s*(.*)$", s)
            if not m:
                continue
            d, rest = m.group(1), m.group(2)
            if d in ("ifdef", "ifndef", "if"):
                if not active:
                    stack.append((False, True, False)); active = False
                    continue
                if d == "if":
                    c = self.cond(tokenize(rest))
                else:
                    c = (rest.split()[0] in self.macros) == (d == "ifdef")
                stack.append((True, c, c)); active = c
            elif d == "elif":
                par, taken, _ = stack[-1]
                c = par and not taken and self.cond(tokenize(rest))
                stack[-1] = (par, taken or c, c); active = c
            elif d == "else":
                par, taken, _ = stack[-1]
                c = par and not taken
                stack[-1] = (par, True, c); active = c
            elif d == "endif":
                stack.pop()
                active = stack[-1][2] if stack else True
            elif not active:
                continue
            elif d == "define":
                mm = re.match(r"(\w+)(\(([^)]*)\))?\s*(.*)$", rest)
                name = mm.group(1)
                if mm.group(2) is not None and rest[len(name):len(name) + 1] == "(":
                    params = [p.strip() for p in mm.group(3).split(",") if p.strip()]
                    self.macros[name] = Macro(params, tokenize(mm.group(4)))
                else:
                    self.macros[name] = Macro(None, tokenize(rest[len(name):]))
            elif d == "undef":
                self.macros.pop(rest.split()[0], None)
        if pending:
            out.extend(self.expand(pending))
        return out


# ---------------------------------------------------------------------------------------------------- parser
# AST nodes are tuples: (kind, ...).  Types: T instance | ("ptr", type) | ("arr", type, n_or_None) | VOID.

_BINPREC = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", ">", "<=", ">="), ("<<", ">>"), ("+", "-"),
            ("*", "/", "%")]
_ASSIGN = {"=", "+=", "-=", "*=", "/=", "%=", "<<=", ">>=", "&=", "|=", "^="}


class Parser:
    def __init__(self, toks, typedefs, consts=None, interp=None, structs=None):
        self.t, self.i, self.typedefs = toks, 0, typedefs
        self.structs = structs if structs is not None else {}
        self.lenient = 0             # > 0 inside parameter lists: unknown type names become opaque (incomplete) types
        self.consts, self.interp = consts, interp     # enum constants are registered in `consts` as they are parsed

    # -- token helpers
    def peek(self, k=0):
        j = self.i + k
        return self.t[j][1] if j < len(self.t) else None

    def kind(self, k=0):
        j = self.i + k
        return self.t[j][0] if j < len(self.t) else None

    def next(self):
        s = self.t[self.i][1]; self.i += 1
        return s

    def accept(self, s):
        if self.peek() == s and self.kind() in ("op", "id"):
            self.i += 1
            return True
        return False

    def expect(self, s):
        if not self.accept(s):
            raise CError("expected %r, got %r (token %d: %s)" % (s, self.peek(), self.i, " ".join(x[1] for x in self.t[max(0, self.i - 8):self.i + 4])))

    def is_type_start(self, k=0):
        s = self.peek(k)
        return self.kind(k) == "id" and (s in TYPE_WORDS or s in QUALIFIERS or s in self.typedefs or s in ("struct", "union", "enum")
                                         or ("<opaque>" + s) in self.structs)

    # -- declarations
    def specifiers(self):
        words = []
        base = None
        while self.kind() == "id":
            s = self.peek()
            if s in QUALIFIERS:
                self.i += 1
            elif s in TYPE_WORDS:
                words.append(s); self.i += 1
            elif s in self.typedefs and base is None and not words:
                base = self.typedefs[s]; self.i += 1
            elif s == "enum" and base is None and not words:
                self.i += 1
                if self.kind() == "id" and self.peek() != "{":
                    self.i += 1                      # tag
                if self.accept("{"):
                    nxt = 0
                    while not self.accept("}"):
                        nm = self.next()
                        if self.accept("="):
                            nxt = (self.interp or Interp({}, {}, {})).ev(self.cond_expr())[0]
                        if self.consts is not None:
                            self.consts[nm] = Ptr([nxt], 0, I32)
                        nxt += 1
                        if not self.accept(","):
                            self.expect("}")
                            break
                base = U32 if False else I32
            elif s == "struct" and base is None and not words:
                self.i += 1
                tag = None
                if self.kind() == "id" and self.peek() != "{":
                    tag = self.next()
                if tag is None:
                    st = StructType("<anon%d>" % self.i)
                else:
                    st = self.structs.get(tag)
                    if st is None:
                        st = self.structs[tag] = StructType(tag)
                if self.accept("{"):
                    fields = []
                    while not self.accept("}"):
                        fb = self.specifiers()
                        while True:
                            nm, ty, params = self.declarator(fb)
                            if self.peek() == ":":
                                raise CError("bit-field in struct %s" % st.name)
                            if params is not None:
                                raise CError("function member")
                            fields.append((nm, ty))
                            if self.accept(";"):
                                break
                            self.expect(",")
                    st.fields = fields
                base = st
            elif s == "union":
                raise CError("union not supported")
            elif s == "__attribute__":
                self.i += 1; self.skip_parens()
            elif ((self.lenient or ("<opaque>" + s) in self.structs) and base is None and not words and s not in self.typedefs
                  and (self.peek(1) in ("*", ")", ",") or self.kind(1) == "id")):
                self.i += 1
                base = self.structs.get("<opaque>" + s)
                if base is None:
                    base = self.structs["<opaque>" + s] = StructType(s)
            else:
                break
        if base is not None:
            return base
        if not words:
            raise CError("type expected at %r" % self.peek())
        w = set(words)
        if "void" in w:
            return VOID
        if "double" in w or "float" in w:
            return F64
        uns = "unsigned" in w
        if "char" in w:
            return U8 if uns else I8
        if "short" in w:
            return U16 if uns else I16
        if "long" in w:
            return U64 if uns else I64
        if "_Bool" in w:
            return U8
        return U32 if uns else I32

    def skip_parens(self):
        self.expect("(")
        depth = 1
        while depth:
            s = self.next()
            if s == "(":
                depth += 1
            elif s == ")":
                depth -= 1

    def declarator(self, base, end=None):
        """Returns (name or None, type, params or None).  `end`: parse a sub-range (nested declarator)."""
        while self.accept("*"):
            base = ("ptr", base)
            while self.peek() in QUALIFIERS:
                self.i += 1
        name, params, inner = None, None, None
        if self.peek() == "(" and (self.peek(1) in ("*", "(") or (self.kind(1) == "id" and self.peek(2) == ")" and not self.is_type_start(1))):
            # nested declarator: remember its token range, apply the suffixes to `base` first
            depth, j = 0, self.i
            while True:
                s = self.t[j][1]
                if s == "(":
                    depth += 1
                elif s == ")":
                    depth -= 1
                    if depth == 0:
                        break
                j += 1
            inner = (self.i + 1, j)
            self.i = j + 1
        elif self.kind() == "id" and self.peek() not in QUALIFIERS:
            name = self.next()
        dims = []
        while True:
            if self.accept("["):
                if self.accept("]"):
                    dims.append(None)
                else:
                    dims.append(self.cond_expr()); self.expect("]")
            elif self.peek() == "(" and inner is None and params is None and not dims:
                params = self.param_list()
            elif self.peek() == "(" and inner is not None:
                self.skip_parens(); base = ("fn", base)
            elif self.peek() == "__attribute__":
                self.i += 1; self.skip_parens()
            else:
                break
        for d in reversed(dims):
            base = ("arr", base, d)
        if inner is not None:
            save = self.i
            self.i = inner[0]
            name, base, _ = self.declarator(base)
            if self.i != inner[1]:
                raise CError("nested declarator")
            self.i = save
        return name, base, params

    def param_list(self):
        self.lenient += 1
        try:
            return self.param_list_()
        finally:
            self.lenient -= 1

    def param_list_(self):
        self.expect("(")
        params = []
        if self.accept(")"):
            return params
        if self.peek() == "void" and self.peek(1) == ")":
            self.i += 2
            return params
        while True:
            if self.accept("..."):
                params.append(("...", None))
            else:
                b = self.specifiers()
                nm, ty, _ = self.declarator(b)
                if isinstance(ty, tuple) and ty[0] == "arr":
                    ty = ("ptr", ty[1])
                params.append((nm, ty))
            if self.accept(")"):
                return params
            self.expect(",")

    def initializer(self):
        if self.accept("{"):
            items = []
            while not self.accept("}"):
                items.append(self.initializer())
                if not self.accept(","):
                    self.expect("}")
                    break
            return ("initlist", items)
        return self.assign_expr()

    def declaration(self):
        base = self.specifiers()
        decls = []
        if self.accept(";"):
            return ("decl", decls)
        while True:
            nm, ty, params = self.declarator(base)
            init = self.initializer() if self.accept("=") else None
            decls.append((nm, ty, init))
            if self.accept(";"):
                return ("decl", decls)
            self.expect(",")

    # -- statements
    def statement(self):
        s = self.peek()
        k = self.kind()
        if s == "{" and k == "op":
            return self.compound()
        if k == "id":
            if s == "if":
                self.i += 1; self.expect("(")
                c = self.expr(); self.expect(")")
                a = self.statement()
                b = self.statement() if self.accept("else") else None
                return ("if", c, a, b)
            if s == "for":
                self.i += 1; self.expect("(")
                if self.accept(";"):
                    init = None
                elif self.is_type_start():
                    init = self.declaration()
                else:
                    init = ("expr", self.expr()); self.expect(";")
                cond = None if self.peek() == ";" else self.expr()
                self.expect(";")
                step = None if self.peek() == ")" else self.expr()
                self.expect(")")
                return ("for", init, cond, step, self.statement())
            if s == "while":
                self.i += 1; self.expect("(")
                c = self.expr(); self.expect(")")
                return ("while", c, self.statement())
            if s == "do":
                self.i += 1
                body = self.statement()
                self.expect("while"); self.expect("(")
                c = self.expr(); self.expect(")"); self.expect(";")
                return ("do", body, c)
            if s == "switch":
                self.i += 1; self.expect("(")
                c = self.expr(); self.expect(")")
                return ("switch", c, self.statement())
            if s == "case":
                self.i += 1
                v = self.cond_expr(); self.expect(":")
                return ("case", v)
            if s == "default" and self.peek(1) == ":":
                self.i += 2
                return ("default",)
            if s == "break":
                self.i += 1; self.expect(";")
                return ("break",)
            if s == "continue":
                self.i += 1; self.expect(";")
                return ("continue",)
            if s == "return":
                self.i += 1
                if self.accept(";"):
                    return ("return", None)
                e = self.expr(); self.expect(";")
                return ("return", e)
            if self.is_type_start():
                return self.declaration()
        if self.accept(";"):
            return ("empty",)
        e = self.expr(); self.expect(";")
        return ("expr", e)

    def compound(self):
        self.expect("{")
        body = []
        while not self.accept("}"):
            body.append(self.statement())
        has_decl = any(b[0] == "decl" for b in body)
        return ("block", body, has_decl)

    # -- expressions
    def expr(self):
        e = self.assign_expr()
        while self.accept(","):
            e = ("comma", e, self.assign_expr())
        return e

    def assign_expr(self):
        lhs = self.cond_expr()
        s = self.peek()
        if s in _ASSIGN and self.kind() == "op":
            self.i += 1
            rhs = self.assign_expr()
            return ("assign", s, lhs, rhs)
        return lhs

    def cond_expr(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.expr(); self.expect(":")
            b = self.cond_expr()
            return ("cond", c, a, b)
        return c

    def binary(self, lvl):
        if lvl == len(_BINPREC):
            return self.unary()
        e = self.binary(lvl + 1)
        ops = _BINPREC[lvl]
        while self.kind() == "op" and self.peek() in ops:
            op = self.next()
            r = self.binary(lvl + 1)
            e = ("bin", op, e, r)
        return e

    def type_name(self):
        b = self.specifiers()
        _, ty, _ = self.declarator(b)
        return ty

    def unary(self):
        s, k = self.peek(), self.kind()
        if k == "op":
            if s in ("-", "+", "!", "~"):
                self.i += 1
                return ("un", s, self.unary())
            if s == "*":
                self.i += 1
                return ("deref", self.unary())
            if s == "&":
                self.i += 1
                return ("addr", self.unary())
            if s in ("++", "--"):
                self.i += 1
                return ("preinc", s, self.unary())
            if s == "(" and self.is_type_start(1):
                self.i += 1
                ty = self.type_name(); self.expect(")")
                return ("cast", ty, self.unary())
        if k == "id" and s == "sizeof":
            self.i += 1
            if self.peek() == "(" and self.is_type_start(1):
                self.i += 1
                ty = self.type_name(); self.expect(")")
                return ("sizeof_t", ty)
            return ("sizeof_e", self.unary())
        return self.postfix()

    def postfix(self):
        k, s = self.kind(), self.peek()
        if k == "num":
            self.i += 1
            e = ("num",) + parse_int_literal(s)
        elif k == "chr":
            self.i += 1
            body = s[1:-1]
            v = {"\\n": 10, "\\0": 0, "\\t": 9, "\\\\": 92, "\\'": 39}.get(body, ord(body[0]) if len(body) == 1 else None)
            e = ("num", v, I32)
        elif k == "str":
            self.i += 1
            e = ("str", s)
        elif k == "id":
            self.i += 1
            e = ("var", s)
        elif s == "(":
            self.i += 1
            e = self.expr(); self.expect(")")
        else:
            raise CError("unexpected token %r" % s)
        while True:
            s = self.peek()
            if self.kind() != "op":
                break
            if s == "[":
                self.i += 1
                ix = self.expr(); self.expect("]")
                e = ("index", e, ix)
            elif s == "(":
                self.i += 1
                args = []
                if not self.accept(")"):
                    while True:
                        args.append(self.assign_expr())
                        if self.accept(")"):
                            break
                        self.expect(",")
                e = ("call", e, args)
            elif s in ("++", "--"):
                self.i += 1
                e = ("postinc", s, e)
            elif s in (".", "->"):
                self.i += 1
                e = ("member", e, self.next(), s == "->")
            else:
                break
        return e


def parse_int_literal(s):
    m = re.match(r"^(0[xX][0-9a-fA-F]+|\d+)([uUlL]*)$", s)
    if not m:
        return float(s.rstrip("fFlL")), F64
    txt, suf = m.group(1), m.group(2).lower()
    hexa = txt[:2].lower() == "0x"
    v = int(txt, 16) if hexa else (int(txt, 8) if len(txt) > 1 and txt[0] == "0" else int(txt))
    uns, lng = "u" in suf, "l" in suf
    cands = []
    if not lng:
        if not uns:
            cands.append(I32)
        if uns or hexa:
            cands.append(U32)
    if not uns:
        cands.append(I64)
    if uns or hexa:
        cands.append(U64)
    for t in cands:
        lo, hi = (-(1 << (t.bits - 1)), (1 << (t.bits - 1)) - 1) if t.signed else (0, t.mask)
        if lo <= v <= hi:
            return v, t
    raise CError("literal out of range: " + s)


# ------------------------------------------------------------------------------------------------ interpreter
# Every object lives in a "storage" Ptr: a variable `int x` is Ptr([v], 0, I32); `int a[8][15]` is
# Ptr(buf, 0, I32, dims=(8, 15)) whose deref() is the decayed pointer-to-row; a struct variable is
# Ptr([StructVal], 0, StructType).  Reading a variable = storage.deref(); `&x` = the storage itself.

class Func:
    __slots__ = ("name", "ret", "params", "toks", "body")


_BRK, _CONT = ("brk",), ("cont",)


class Interp:
    def __init__(self, funcs, globs, typedefs, structs=None):
        self.funcs, self.globs, self.typedefs = funcs, globs, typedefs
        self.structs = structs if structs is not None else {}
        self.scopes = []
        self.protos = set()        # functions declared but not (yet) defined: usable as function-pointer values
        self.pycalls = {}          # name -> python callable(interp, [(v, tag), ...]) -> (v, tag): host-provided functions

    # -- types and storage
    def sizeof(self, ty):
        c = ty.__class__
        if c is T:
            return ty.size
        if c is StructType:
            if ty.fields is None:
                raise CError("sizeof incomplete %s" % ty)
            return sum(self.sizeof(ft) for _, ft in ty.fields)
        if ty[0] == "ptr":
            return 8
        if ty[0] == "arr":
            d = ty[2]
            return self.sizeof(ty[1]) * (d if isinstance(d, int) else self.ev(d)[0])
        raise CError("sizeof(%r)" % (ty,))

    def flatten(self, ty, init=None):
        """("arr", ("arr", e, m), n) -> ([n, m], e); an unsized first dimension is taken from the initialiser."""
        dims = []
        while ty.__class__ is tuple and ty[0] == "arr":
            d = ty[2]
            dims.append(None if d is None else (d if isinstance(d, int) else self.ev(d)[0]))
            ty = ty[1]
        if dims and dims[0] is None:
            if init is None or init[0] != "initlist":
                raise CError("unsized array without initialiser")
            items = init[1]
            if len(dims) == 1 and ty.__class__ is not StructType or all(i[0] == "initlist" for i in items):
                dims[0] = len(items)
            else:
                sub = 1
                for d in dims[1:]:
                    sub *= d
                per = sub * (len(ty.fields) if ty.__class__ is StructType else 1)
                dims[0] = -(-len(items) // per)
        return dims, ty

    def blank(self, ty, zero):
        c = ty.__class__
        if c is T:
            return 0 if zero else None
        if c is StructType:
            if ty.fields is None:
                raise CError("object of incomplete %s" % ty)
            return StructVal(ty, {fn: self.alloc(ft, zero) for fn, ft in ty.fields})
        return None if zero else UNINIT

    def alloc(self, ty, zero=False, init=None):
        if ty.__class__ is tuple and ty[0] == "arr":
            dims, et = self.flatten(ty, init)
            n = 1
            for d in dims:
                n *= d
            if et.__class__ is StructType:
                buf = [self.blank(et, zero) for _ in range(n)]
            else:
                buf = [self.blank(et, zero)] * n
            return Ptr(buf, 0, et, tuple(dims))
        if ty.__class__ is tuple and ty[0] == "fn":
            raise CError("object of function type")
        return Ptr([self.blank(ty, zero)], 0, ty)

    def zero_fill(self, st):
        """Zero every scalar reachable from storage `st` (aggregate initialisers zero what they do not name)."""
        n = st.stride * (st.dims[0] if st.dims else 1) if st.dims else 1
        n = 1
        for d in st.dims:
            n *= d
        for i in range(st.off, st.off + n):
            v = st.buf[i]
            if st.t.__class__ is T:
                st.buf[i] = 0
            elif st.t.__class__ is StructType:
                for f in v.f.values():
                    self.zero_fill(f)
            else:
                st.buf[i] = None

    def init_storage(self, st, init):
        """Run a C initialiser on freshly allocated storage `st` (scalar, array or struct)."""
        if init[0] != "initlist":
            v, vt = self.ev(init)
            if st.dims:
                raise CError("array initialised from an expression")
            st.store(v, vt)
            return
        self.zero_fill(st)
        it = _Items(init[1])
        self.consume(st, it)
        if it.pos < len(it.items):
            raise CError("too many initialisers")

    def consume(self, st, it):
        """Initialise the object at `st` from the item stream (C brace-elision rules)."""
        if st.dims:
            sub = Ptr(st.buf, st.off, st.t, st.dims[1:])
            for k in range(st.dims[0]):
                if it.pos >= len(it.items):
                    return
                elem = sub.add(k)
                nxt = it.items[it.pos]
                if nxt[0] == "initlist" and (elem.dims or elem.t.__class__ is StructType):
                    it.pos += 1
                    inner = _Items(nxt[1])
                    self.consume(elem, inner)
                else:
                    self.consume(elem, it)
            return
        if st.t.__class__ is StructType:
            nxt = it.items[it.pos]
            if nxt[0] != "initlist":
                # an expression of the struct type initialises the whole member; otherwise braces are elided
                try:
                    tag = self.static_type(nxt)
                except CError:
                    tag = None
                if tag is st.t:
                    it.pos += 1
                    v, vt = self.ev(nxt)
                    st.store(v, vt)
                    return
            sv = st.buf[st.off]
            for fn, _ in st.t.fields:
                if it.pos >= len(it.items):
                    return
                fst = sv.f[fn]
                nxt = it.items[it.pos]
                if nxt[0] == "initlist" and (fst.dims or fst.t.__class__ is StructType):
                    it.pos += 1
                    self.consume(fst, _Items(nxt[1]))
                else:
                    self.consume(fst, it)
            return
        nxt = it.items[it.pos]
        it.pos += 1
        if nxt[0] == "initlist":
            if len(nxt[1]) != 1:
                raise CError("braced scalar initialiser")
            nxt = nxt[1][0]
        v, vt = self.ev(nxt)
        st.store(v, vt)

    def declare(self, scope, name, ty, init):
        if ty is VOID:
            raise CError("void object %s" % name)
        st = self.alloc(ty, False, init)
        if init is not None:
            self.init_storage(st, init)
        scope[name] = st

    def lookup(self, name):
        for sc in reversed(self.scopes):
            v = sc.get(name)
            if v is not None:
                return v
        v = self.globs.get(name)
        if v is None:
            raise CError("unknown identifier %s" % name)
        return v

    @staticmethod
    def const_eval(node):
        return Interp({}, {}, {}).ev(node)

    # -- lvalues: the storage Ptr of the designated object
    def lv(self, node):
        k = node[0]
        if k == "var":
            st = self.lookup(node[1])
            if st.dims:
                raise CError("array %s is not assignable" % node[1])
            return st
        if k == "index":
            p = self.ptr_of(("bin", "+", node[1], node[2]))
        elif k == "deref":
            p = self.ptr_of(node[1])
        elif k == "member":
            return self.member_storage(node)
        elif k == "cast":
            return self.lv(node[2])
        else:
            raise CError("not an lvalue: %s" % k)
        if p.dims:
            raise CError("lvalue is an array")
        return p

    def member_storage(self, node):
        if node[3]:
            p = self.ptr_of(node[1])
            sv, t = p.deref()
        else:
            sv, t = self.ev(node[1])
        if t.__class__ is not StructType:
            raise CError("member access on non-struct")
        st = sv.f.get(node[2])
        if st is None:
            raise CError("%s has no member %s" % (t, node[2]))
        return st

    def ptr_of(self, node):
        v, t = self.ev(node)
        if t is not PTR or v.__class__ is not Ptr:
            raise CError("null or non-pointer dereference")
        return v

    # -- expressions
    def arith(self, op, a, at, b, bt):
        if at is PTR or bt is PTR:
            return self.ptr_arith(op, a, at, b, bt)
        if at.__class__ is not T or bt.__class__ is not T:
            raise CError("arithmetic on non-scalar")
        if op == "<<" or op == ">>":
            t = at if at.bits >= 32 else I32
            a = wrap(a, t)
            if not 0 <= b < t.bits:
                raise CError("shift count %d out of range" % b)
            return (wrap(a << b, t) if op == "<<" else a >> b), t
        t = common(at, bt)
        if t is F64:
            a, b = float(a), float(b)
            if op == "+":
                return a + b, t
            if op == "-":
                return a - b, t
            if op == "*":
                return a * b, t
            if op == "/":
                return a / b, t
            if op in ("<", ">", "<=", ">=", "==", "!="):
                return int({"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b, "==": a == b, "!=": a != b}[op]), I32
            raise CError("operator %s on double" % op)
        a, b = wrap(a, t), wrap(b, t)
        if op == "+":
            return wrap(a + b, t), t
        if op == "-":
            return wrap(a - b, t), t
        if op == "*":
            return wrap(a * b, t), t
        if op == "<":
            return int(a < b), I32
        if op == ">":
            return int(a > b), I32
        if op == "<=":
            return int(a <= b), I32
        if op == ">=":
            return int(a >= b), I32
        if op == "==":
            return int(a == b), I32
        if op == "!=":
            return int(a != b), I32
        if op == "&":
            return wrap(a & b, t), t
        if op == "|":
            return wrap(a | b, t), t
        if op == "^":
            return wrap(a ^ b, t), t
        if op == "/" or op == "%":
            if b == 0:
                raise CError("division by zero")
            q = abs(a) // abs(b)
            if (a < 0) != (b < 0):
                q = -q
            return (wrap(q, t), t) if op == "/" else (wrap(a - q * b, t), t)
        raise CError("operator " + op)

    def ptr_arith(self, op, a, at, b, bt):
        if op == "+":
            if at is PTR and bt is not PTR:
                return a.add(b), PTR
            if bt is PTR and at is not PTR:
                return b.add(a), PTR
        elif op == "-":
            if at is PTR and bt is not PTR:
                return a.add(-b), PTR
            if at is PTR and bt is PTR:
                if a.buf is not b.buf:
                    raise CError("difference of unrelated pointers")
                return (a.off - b.off) // a.stride, I64
        elif op in ("==", "!=", "<", ">", "<=", ">="):
            def key(x, xt):
                if xt is not PTR:
                    if x != 0:
                        raise CError("pointer compared with integer")
                    return None
                if x is None:
                    return None
                return (id(x.buf), x.off) if x.__class__ is Ptr else (id(x), 0)
            ka, kb = key(a, at), key(b, bt)
            if op == "==":
                return int(ka == kb), I32
            if op == "!=":
                return int(ka != kb), I32
            if ka is None or kb is None or ka[0] != kb[0]:
                raise CError("ordering of unrelated pointers")
            return int({"<": ka[1] < kb[1], ">": ka[1] > kb[1], "<=": ka[1] <= kb[1], ">=": ka[1] >= kb[1]}[op]), I32
        raise CError("pointer operator " + op)

    def truth(self, node):
        v, t = self.ev(node)
        return v is not None if t is PTR else v != 0

    def ev(self, node):
        k = node[0]
        if k == "num":
            return node[1], node[2]
        if k == "var":
            name = node[1]
            for sc in reversed(self.scopes):
                st = sc.get(name)
                if st is not None:
                    return st.deref()
            st = self.globs.get(name)
            if st is not None:
                return st.deref()
            if name in self.funcs or name in self.pycalls or name in self.protos:
                return FuncRef(name), PTR
            if name + "_c" in self.funcs:       # rtcd name used as a function-pointer value (generic target: the _c symbol)
                return FuncRef(name + "_c"), PTR
            raise CError("unknown identifier %s" % name)
        if k == "bin":
            op = node[1]
            if op == "&&":
                return int(self.truth(node[2]) and self.truth(node[3])), I32
            if op == "||":
                return int(self.truth(node[2]) or self.truth(node[3])), I32
            a, at = self.ev(node[2])
            b, bt = self.ev(node[3])
            return self.arith(op, a, at, b, bt)
        if k == "index":
            a, at = self.ev(node[1])
            b, bt = self.ev(node[2])
            if at is not PTR:
                a, at, b, bt = b, bt, a, at
            if at is not PTR or a.__class__ is not Ptr:
                raise CError("indexing a non-pointer")
            return a.add(b).deref()
        if k == "member":
            return self.member_storage(node).deref()
        if k == "assign":
            ref = self.lv(node[2])
            if node[1] == "=":
                v, vt = self.ev(node[3])
                return ref.store(v, vt)
            a, at = ref.deref()
            b, bt = self.ev(node[3])
            v, vt = self.arith(node[1][:-1], a, at, b, bt)
            return ref.store(v, vt)
        if k == "call":
            return self.call_node(node)
        if k == "un":
            op = node[1]
            if op == "!":
                return int(not self.truth(node[2])), I32
            v, t = self.ev(node[2])
            if t.__class__ is not T:
                raise CError("unary %s on non-integer" % op)
            t = promote(t)
            if op == "-":
                return wrap(-v, t), t
            if op == "~":
                return wrap(~v, t), t
            return wrap(v, t), t
        if k == "cond":
            if self.truth(node[1]):
                v, t = self.ev(node[2])
                other = node[3]
            else:
                v, t = self.ev(node[3])
                other = node[2]
            if t.__class__ is not T:
                return v, t
            ot = self.static_type(other)
            if ot is None:
                return wrap(v, promote(t)), promote(t)
            if ot.__class__ is not T:
                return v, t
            ct = common(t, ot)
            return wrap(v, ct), ct
        if k == "cast":
            ty = node[1]
            v, t = self.ev(node[2])
            if ty is VOID:
                return None, VOID
            if ty.__class__ is T:
                if t.__class__ is not T:
                    raise CError("pointer to integer cast")
                return wrap(v, ty), ty
            if ty.__class__ is StructType:
                if t is not ty:
                    raise CError("cast to struct")
                return v, t
            if t is PTR:
                # pointer casts keep the buffer; a cast between integer types of one size gives a re-typed view
                if v.__class__ is Ptr and ty[0] == "ptr" and ty[1].__class__ is T and v.t.__class__ is T and v.dims:
                    # (int16_t *)table2d: a pointer to rows becomes a pointer to the first scalar of that row
                    if ty[1].size != v.t.size:
                        raise CError("pointer cast changes the element size (%s -> %s)" % (v.t, ty[1]))
                    return Ptr(v.buf, v.off, ty[1] if ty[1] is not v.t else v.t), PTR
                if (v.__class__ is Ptr and ty[0] == "ptr" and ty[1].__class__ is T and v.t.__class__ is T and ty[1] is not v.t
                        and not v.dims):
                    if ty[1] is U8 and v.t is U16:
                        # libaom's byte-pointer encoding of a high-bit-depth buffer (`(uint8_t *)p16`, aom_ports/mem.h:79-80):
                        # such a pointer is only ever handed to CONVERT_TO_SHORTPTR, which is the identity in this
                        # (buffer, element) pointer model, so it keeps its real element type
                        return v, PTR
                    if ty[1].size != v.t.size:
                        raise CError("pointer cast changes the element size (%s -> %s)" % (v.t, ty[1]))
                    return Ptr(v.buf, v.off, ty[1]), PTR
                return v, PTR
            if t.__class__ is T and v == 0:
                return None, PTR
            raise CError("integer to pointer cast")
        if k == "deref":
            v, t = self.ev(node[1])
            if t is PTR and v.__class__ is FuncRef:
                return v, PTR
            if t is not PTR or v.__class__ is not Ptr:
                raise CError("null or non-pointer dereference")
            return v.deref()
        if k == "addr":
            inner = node[1]
            ik = inner[0]
            if ik == "var":
                name = inner[1]
                try:
                    st = self.lookup(name)
                except CError:
                    if name in self.funcs or name in self.pycalls:
                        return FuncRef(name), PTR
                    raise
                return (Ptr(st.buf, st.off, st.t, st.dims[1:]) if st.dims else st), PTR
            if ik == "index":
                return self.ev(("bin", "+", inner[1], inner[2]))
            if ik == "deref":
                return self.ev(inner[1])
            if ik == "member":
                st = self.member_storage(inner)
                return (Ptr(st.buf, st.off, st.t, st.dims[1:]) if st.dims else st), PTR
            raise CError("address-of")
        if k == "preinc" or k == "postinc":
            ref = self.lv(node[2])
            a, at = ref.deref()
            d = 1 if node[1] == "++" else -1
            if at is PTR:
                ref.store(a.add(d), PTR)
                return (a.add(d) if k == "preinc" else a), PTR
            nv, nt = ref.store(a + d, promote(at))
            return (nv, nt) if k == "preinc" else (a, at)
        if k == "comma":
            self.ev(node[1])
            return self.ev(node[2])
        if k == "sizeof_t":
            return self.sizeof(node[1]), U64
        if k == "sizeof_e":
            inner = node[1]
            if inner[0] == "var" or inner[0] == "member":     # sizeof an object (arrays do not decay here)
                st = self.lookup(inner[1]) if inner[0] == "var" else self.member_storage(inner)
                n = 1
                for d in st.dims:
                    n *= d
                return n * self.sizeof(st.t), U64
            if inner[0] == "deref" or inner[0] == "index":    # sizeof(*p), sizeof(p[i]): the operand is not evaluated
                pv = self.ev(inner[1])[0] if inner[0] == "deref" else self.ev(("bin", "+", inner[1], inner[2]))[0]
                if pv.__class__ is not Ptr:
                    raise CError("sizeof through a non-pointer")
                return self.sizeof(pv.t) * (pv.stride if pv.dims else 1), U64
            v, t = self.ev(inner)
            if t is PTR:
                return (self.sizeof(v.t) * v.stride if v.__class__ is Ptr and v.dims else 8), U64
            return self.sizeof(t), U64
        if k == "str":
            return node[1], "str"
        raise CError("cannot evaluate node %s" % k)

    def tag_of(self, ty):
        c = ty.__class__
        if c is T or c is StructType:
            return ty
        return VOID if ty is VOID else PTR

    def static_type(self, node):
        """Type tag an expression would have, without evaluating it (needed for the untaken arm of ?:); None = unknown."""
        k = node[0]
        if k == "num":
            return node[2]
        if k == "var":
            try:
                st = self.lookup(node[1])
            except CError:
                return PTR
            return PTR if st.dims else self.tag_of(st.t)
        if k == "cast":
            return self.tag_of(node[1])
        if k == "bin":
            op = node[1]
            if op in ("&&", "||", "<", ">", "<=", ">=", "==", "!="):
                return I32
            a = self.static_type(node[2])
            if a is None:
                return None
            if op in ("<<", ">>"):
                return PTR if a is PTR else promote(a)
            b = self.static_type(node[3])
            if b is None:
                return None
            if a is PTR or b is PTR:
                return I64 if (a is PTR and b is PTR) else PTR
            return common(a, b)
        if k == "un":
            if node[1] == "!":
                return I32
            a = self.static_type(node[2])
            return None if a is None else promote(a)
        if k == "cond":
            a, b = self.static_type(node[2]), self.static_type(node[3])
            if a is None or b is None:
                return a if b is None else b
            if a.__class__ is not T or b.__class__ is not T:
                return a
            return common(a, b)
        if k in ("index", "deref", "member"):
            try:      # element types are dynamic: evaluate the address (side-effect free in the evaluated code)
                if k == "index":
                    p = self.ev(("bin", "+", node[1], node[2]))[0]
                elif k == "deref":
                    p = self.ev(node[1])[0]
                else:
                    p = self.member_storage(node)
            except (CError, AttributeError):
                return None
            if p.__class__ is not Ptr:
                return None
            return PTR if p.dims else self.tag_of(p.t)
        if k == "call":
            f = node[1][1] if node[1][0] == "var" else None
            if f in self.funcs:
                return self.tag_of(self.funcs[f].ret)
            return I32
        if k == "assign" or k in ("preinc", "postinc"):
            return self.static_type(node[2])
        if k == "comma":
            return self.static_type(node[2])
        if k in ("sizeof_t", "sizeof_e"):
            return U64
        if k == "addr":
            return PTR
        raise CError("static type of %s" % k)

    # -- calls
    def call_node(self, node):
        callee = node[1]
        args = [self.ev(a) for a in node[2]]
        if callee[0] == "var":
            name = callee[1]
            shadow = False
            for sc in self.scopes:
                if name in sc:
                    shadow = True
            if not shadow and name not in self.globs:
                if name in self.funcs:
                    return self.call(name, args)
                if name in self.pycalls:
                    return self.pycalls[name](self, args)
                if name + "_c" in self.funcs:
                    # rtcd dispatch of the generic target: every `aom_foo` is `#define aom_foo aom_foo_c` in the generated
                    # config/*_rtcd.h (build/cmake/rtcd.pl:144-166 with a single implementation)
                    return self.call(name + "_c", args)
                return self.builtin(name, args)
        f, ft = self.ev(callee)
        if ft is not PTR or f is None:
            raise CError("call through a null / non-function value")
        if f.__class__ is FuncRef:
            if f.name in self.funcs:
                return self.call(f.name, args)
            return self.pycalls[f.name](self, args)
        if callable(f):
            return f(self, args)
        raise CError("call through a data pointer")

    def builtin(self, name, args):
        if name in ("abs", "labs", "llabs"):
            v, t = args[0]
            t = I32 if name == "abs" else I64
            v = wrap(v, t)
            return wrap(-v if v < 0 else v, t), t
        if name == "assert":
            if args[0][0] is None or args[0][0] == 0:
                raise CError("assert failed")
            return None, VOID
        if name == "memset":
            (p, _), (c, _), (n, _) = args
            if p.t.__class__ is not T:
                raise CError("memset on non-integer elements")
            cnt, rem = divmod(n, p.t.size)
            if rem:
                raise CError("memset size not a multiple of the element size")
            val = wrap(int.from_bytes(bytes([c & 0xff]) * p.t.size, "little"), p.t)
            if p.off < 0 or p.off + cnt > len(p.buf):
                raise CError("memset out of bounds")
            for i in range(p.off, p.off + cnt):
                p.buf[i] = val
            return p, PTR
        if name in ("memcpy", "memmove"):
            (d, _), (s, _), (n, _) = args
            if d.t.__class__ is not T or s.t.__class__ is not T or d.t.size != s.t.size:
                raise CError("memcpy between different element kinds")
            cnt, rem = divmod(n, d.t.size)
            if rem:
                raise CError("memcpy size not a multiple of the element size")
            if s.off < 0 or s.off + cnt > len(s.buf) or d.off < 0 or d.off + cnt > len(d.buf):
                raise CError("memcpy out of bounds")
            tmp = s.buf[s.off:s.off + cnt]
            d.buf[d.off:d.off + cnt] = [None if v is None else wrap(v, d.t) for v in tmp]
            return d, PTR
        if name == "__builtin_clz":
            v = wrap(args[0][0], U32)
            if v == 0:
                raise CError("clz(0)")
            return 32 - v.bit_length(), I32
        if name == "__builtin_ctz":
            v = wrap(args[0][0], U32)
            if v == 0:
                raise CError("ctz(0)")
            return (v & -v).bit_length() - 1, I32
        if name in ("printf", "fprintf"):
            return 0, I32
        raise CError("call to unknown function %s" % name)

    def call(self, name, args):
        f = self.funcs[name]
        if f.body is None:
            p = Parser(f.toks, self.typedefs, self.globs, self, self.structs)
            f.body = p.compound()
        if len(args) != len(f.params):
            raise CError("%s: %d arguments for %d parameters" % (name, len(args), len(f.params)))
        scope = {}
        for (pn, pt), (v, vt) in zip(f.params, args):
            st = self.alloc(pt)
            try:
                st.store(v, vt)
            except CError as e:
                raise CError("%s: parameter %s: %s" % (name, pn, e))
            scope[pn] = st
        saved = self.scopes
        self.scopes = [scope]
        try:
            r = self.ex(f.body)
        finally:
            self.scopes = saved
        if r is not None and r[0] == "ret":
            v, t = r[1]
            rt = f.ret
            if rt.__class__ is T:
                if t.__class__ is not T:
                    raise CError("%s returns a non-integer" % name)
                return wrap(v, rt), rt
            if rt is VOID:
                return None, VOID
            if rt.__class__ is StructType:
                return copy_struct(v), rt
            if t.__class__ is T:   # `return cond ? ptr : NULL;` taking the NULL arm: an integer 0 converted to the pointer return type
                if v != 0:
                    raise CError("%s returns a non-zero integer as a pointer" % name)
                v = None
            return v, PTR
        return None, VOID

    # -- statements: return None, _BRK, _CONT or ("ret", (v, t))
    def ex(self, st):
        k = st[0]
        if k == "expr":
            self.ev(st[1])
            return None
        if k == "block":
            if st[2]:
                self.scopes.append({})
            try:
                for s in st[1]:
                    r = self.ex(s)
                    if r is not None:
                        return r
            finally:
                if st[2]:
                    self.scopes.pop()
            return None
        if k == "decl":
            sc = self.scopes[-1]
            for nm, ty, init in st[1]:
                self.declare(sc, nm, ty, init)
            return None
        if k == "if":
            if self.truth(st[1]):
                return self.ex(st[2])
            if st[3] is not None:
                return self.ex(st[3])
            return None
        if k == "for":
            self.scopes.append({})
            try:
                if st[1] is not None:
                    self.ex(st[1])
                while st[2] is None or self.truth(st[2]):
                    r = self.ex(st[4])
                    if r is not None:
                        if r is _BRK:
                            break
                        if r is not _CONT:
                            return r
                    if st[3] is not None:
                        self.ev(st[3])
            finally:
                self.scopes.pop()
            return None
        if k == "while":
            while self.truth(st[1]):
                r = self.ex(st[2])
                if r is not None:
                    if r is _BRK:
                        break
                    if r is not _CONT:
                        return r
            return None
        if k == "do":
            while True:
                r = self.ex(st[1])
                if r is not None:
                    if r is _BRK:
                        break
                    if r is not _CONT:
                        return r
                if not self.truth(st[2]):
                    break
            return None
        if k == "return":
            return ("ret", self.ev(st[1]) if st[1] is not None else (None, VOID))
        if k == "break":
            return _BRK
        if k == "continue":
            return _CONT
        if k == "switch":
            v, _ = self.ev(st[1])
            body = st[2]
            if body[0] != "block":
                raise CError("switch body")
            stmts = body[1]
            start = None
            for i, s in enumerate(stmts):
                if s[0] == "case" and self.ev(s[1])[0] == v:
                    start = i
                    break
            if start is None:
                for i, s in enumerate(stmts):
                    if s[0] == "default":
                        start = i
                        break
            if start is None:
                return None
            self.scopes.append({})
            try:
                for s in stmts[start:]:
                    if s[0] in ("case", "default"):
                        continue
                    r = self.ex(s)
                    if r is not None:
                        if r is _BRK:
                            return None
                        return r
            finally:
                self.scopes.pop()
            return None
        if k in ("empty", "case", "default"):
            return None
        raise CError("statement %s" % k)


class _Items:
    __slots__ = ("items", "pos")

    def __init__(self, items):
        self.items, self.pos = items, 0


# ------------------------------------------------------------------------------------------------- front end

class CEval:
    """Load reference files, then call their functions.

        ev = CEval({"CONFIG_AV1_HIGHBITDEPTH": 1})
        ev.load("/root/reference/aom_ports/mem.h"); ev.load("/root/reference/aom_dsp/quantize.c")
        out = ev.array([0] * 16, "int32_t")
        ev.call("aom_quantize_b_helper_c", coeffs, 16, ..., None, None, 0)
    """

    def __init__(self, defines=None):
        d = {"__GNUC__": 9, "__GNUC_MINOR__": 4, "INLINE": "inline", "NULL": "0", "INT16_MIN": "(-32767-1)", "INT16_MAX": "32767",
             "INT8_MIN": "(-128)", "INT8_MAX": "127", "UINT8_MAX": "255", "UINT16_MAX": "65535", "INT32_MAX": "2147483647",
             "INT32_MIN": "(-2147483647-1)", "INT_MAX": "2147483647", "INT_MIN": "(-2147483647-1)", "UINT32_MAX": "4294967295U",
             "UINT_MAX": "4294967295U", "INT64_MAX": "9223372036854775807L", "INT64_MIN": "(-9223372036854775807L-1)",
             "UINT64_MAX": "18446744073709551615UL"}
        d.update(defines or {})
        self.pp = Preprocessor(d)
        self.typedefs = dict(BASE_TYPEDEFS)
        self.structs = {}
        self.funcs, self.globs = {}, {}
        self.interp = Interp(self.funcs, self.globs, self.typedefs, self.structs)
        self.skipped = []

    def load(self, path):
        with open(path) as fh:
            toks = self.pp.run(fh.read())
        self.toplevel(toks, path)

    def load_text(self, text, name="<text>"):
        self.toplevel(self.pp.run(text), name)

    def define(self, name, body, params=None):
        """(Re)define a macro after loading headers, e.g. the identity for the byte-pointer encoding."""
        self.pp.macros[name] = Macro(params, tokenize(body))

    def toplevel(self, toks, path):
        p = Parser(toks, self.typedefs, self.globs, self.interp, self.structs)
        n = len(toks)
        while p.i < n:
            start = p.i
            try:
                self.top_item(p)
            except (CError, IndexError) as e:
                # skip the construct: to the next ';' at depth 0, or over a '{...}' body
                p.i = start
                depth = 0
                while p.i < n:
                    s = p.next()
                    if s in "{([" and len(s) == 1:
                        depth += 1
                    elif s in "})]" and len(s) == 1:
                        depth -= 1
                        if depth == 0 and s == "}":
                            if p.peek() == ";":
                                p.i += 1
                                break
                            if toks[start][1] not in ("typedef", "struct", "enum", "union"):
                                break                 # function body; `} NAME;` continues a typedef
                    elif s == ";" and depth == 0:
                        break
                self.skipped.append((path, " ".join(x[1] for x in toks[start:start + 8]), str(e)))

    def top_item(self, p):
        if p.accept(";"):
            return
        if p.peek() == "typedef":
            p.i += 1
            base = p.specifiers()
            while True:
                nm, ty, params = p.declarator(base)
                if params is not None:
                    ty = ("fn", ty)
                self.typedefs[nm] = ty
                if p.accept(";"):
                    return
                p.expect(",")
        if p.peek() == "extern" and p.kind(1) == "str":
            p.i += 2
            p.accept("{")
            return
        if p.peek() == "}":          # closing brace of extern "C" {
            p.i += 1
            return
        base = p.specifiers()
        if p.accept(";"):
            return
        while True:
            nm, ty, params = p.declarator(base)
            if params is not None and p.peek() == "{":
                f = Func()
                f.name, f.ret, f.params = nm, ty, params
                depth, j = 0, p.i
                while True:
                    s = p.t[j]
                    if s[0] == "op" and s[1] == "{":
                        depth += 1
                    elif s[0] == "op" and s[1] == "}":
                        depth -= 1
                        if depth == 0:
                            break
                    j += 1
                f.toks, f.body = p.t[p.i:j + 1], None
                p.i = j + 1
                self.funcs[nm] = f
                return
            if params is not None:
                self.interp.protos.add(nm)
            else:
                init = p.initializer() if p.accept("=") else None
                if init is not None or (ty.__class__ is tuple and ty[0] == "arr" and ty[2] is not None) or ty.__class__ is StructType:
                    self.interp.scopes = []
                    if init is None and nm in self.globs:
                        pass
                    else:
                        st = self.interp.alloc(ty, True, init)
                        if init is not None:
                            self.interp.init_storage(st, init)
                        self.globs[nm] = st
            if p.accept(";"):
                return
            p.expect(",")

    # -- Python-side helpers
    def ctype(self, name):
        if not isinstance(name, str):
            return name
        basic = {"int": I32, "unsigned int": U32, "unsigned": U32, "char": I8, "unsigned char": U8, "short": I16,
                 "unsigned short": U16, "long": I64, "unsigned long": U64}
        return basic[name] if name in basic else self.typedefs[name]

    def array(self, values, ctype, dims=()):
        t = self.ctype(ctype)
        return Ptr([wrap(int(v), t) for v in values], 0, t, tuple(dims))

    def new(self, type_name, zero=True):
        """Zero-initialised object of a typedef'd (struct) type; returns its storage pointer."""
        ty = self.typedefs.get(type_name) or self.structs[type_name]
        return self.interp.alloc(ty, zero)

    def field(self, obj, path):
        """Storage of `obj.path` (obj: storage pointer of a struct; path like "mv_limits.row_min" or "site[3]")."""
        st = obj
        for part in re.findall(r"[A-Za-z_]\w*|\[\d+\]", path):
            if part[0] == "[":
                k = int(part[1:-1])
                st = Ptr(st.buf, st.off, st.t, st.dims[1:]).add(k) if st.dims else st.add(k)
            else:
                st = st.buf[st.off].f[part]
        return st

    def set(self, obj, path, value):
        st = self.field(obj, path)
        if value is None or isinstance(value, (Ptr, FuncRef)) or callable(value):
            st.store(value, PTR)
        else:
            st.store(int(value), I64)

    def get(self, obj, path):
        st = self.field(obj, path)
        return st.deref()[0]

    def call(self, name, *args):
        conv = []
        for a in args:
            if a is None or isinstance(a, (Ptr, FuncRef)):
                conv.append((a, PTR))
            elif isinstance(a, StructVal):
                conv.append((a, a.st))
            else:
                conv.append((int(a), I64))
        v, t = self.interp.call(name, conv)
        return v

    def global_values(self, name):
        st = self.globs[name]
        return list(st.buf) if st.dims else st.buf[0]
