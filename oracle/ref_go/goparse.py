"""goparse.py — a lexer and recursive-descent parser for the subset of Go the reference's encoder sources use.

TEST INFRASTRUCTURE (oracle/): part of the build-time translation of the reference's pure-Go zstd encoder into C++ (go2cpp.py),
the way ref_s2asm/plan9_to_gas.py re-spells its assembly.  Nothing of the reference is stored in this repository: the translator
reads /root/reference at build time and writes under oracle/_ref/ (git-ignored).

The AST is plain tuples: (kind, ...).  Types and expressions share the node space (a conversion looks like a call)."""
import re

KEYWORDS = {"break", "case", "chan", "const", "continue", "default", "defer", "else", "fallthrough", "for", "func", "go", "goto",
            "if", "import", "interface", "map", "package", "range", "return", "select", "struct", "switch", "type", "var"}
OPS = ["<<=", ">>=", "&^=", "...", "&&", "||", "<-", "++", "--", "==", "!=", "<=", ">=", ":=", "+=", "-=", "*=", "/=", "%=", "&=", "|=",
       "^=", "<<", ">>", "&^", "+", "-", "*", "/", "%", "&", "|", "^", "<", ">", "=", "!", "(", ")", "[", "]", "{", "}", ",", ";", ".", ":", "~"]
OPS.sort(key=len, reverse=True)
_tok_re = re.compile(r"""
    (?P<ws>[ \t\r]+) |
    (?P<nl>\n) |
    (?P<lc>//[^\n]*) |
    (?P<bc>/\*.*?\*/) |
    (?P<float>(?:\d[\d_]*\.\d[\d_]*(?:[eE][+-]?\d+)?|\d[\d_]*[eE][+-]?\d+|\.\d+(?:[eE][+-]?\d+)?)) |
    (?P<int>0[xX][0-9a-fA-F_]+|0[bB][01_]+|0[oO]?[0-7_]*|[1-9][\d_]*) |
    (?P<id>[^\W\d]\w*) |
    (?P<rune>'(?:\\.[^']*|[^'\\])') |
    (?P<str>"(?:\\.|[^"\\])*") |
    (?P<raw>`[^`]*`) |
    (?P<op>""" + "|".join(re.escape(o) for o in OPS) + r""")
""", re.X | re.S)


class Tok:
    __slots__ = ("kind", "val", "line")

    def __init__(self, kind, val, line):
        self.kind, self.val, self.line = kind, val, line

    def __repr__(self):
        return "%s(%r)@%d" % (self.kind, self.val, self.line)


def lex(src):
    toks, pos, line = [], 0, 1

    def semi_needed():
        if not toks:
            return False
        t = toks[-1]
        if t.kind in ("id", "int", "float", "rune", "str"):
            return True
        if t.kind == "kw":
            return t.val in ("break", "continue", "fallthrough", "return")
        return t.kind == "op" and t.val in ("++", "--", ")", "]", "}")

    while pos < len(src):
        m = _tok_re.match(src, pos)
        if not m:
            raise SyntaxError("lex error at line %d: %r" % (line, src[pos:pos + 30]))
        pos = m.end()
        k = m.lastgroup
        v = m.group(k)
        if k == "ws":
            continue
        if k == "nl" or k == "lc":
            if k == "nl" or True:
                if semi_needed():
                    toks.append(Tok("op", ";", line))
            if k == "nl":
                line += 1
            continue
        if k == "bc":
            if "\n" in v and semi_needed():
                toks.append(Tok("op", ";", line))
            line += v.count("\n")
            continue
        if k == "id" and v in KEYWORDS:
            toks.append(Tok("kw", v, line))
        elif k == "raw":
            toks.append(Tok("str", v, line))
            line += v.count("\n")
        else:
            toks.append(Tok(k, v, line))
    if semi_needed():
        toks.append(Tok("op", ";", line))
    toks.append(Tok("eof", "", line))
    return toks


BINPREC = {"||": 1, "&&": 2, "==": 3, "!=": 3, "<": 3, "<=": 3, ">": 3, ">=": 3, "+": 4, "-": 4, "|": 4, "^": 4,
           "*": 5, "/": 5, "%": 5, "<<": 5, ">>": 5, "&": 5, "&^": 5}


class Parser:
    def __init__(self, src, fname="?"):
        self.toks = lex(src)
        self.i = 0
        self.fname = fname
        self.exprlev = 0  # < 0: composite literals of bare type names are not allowed (statement headers)

    # ---- token helpers ----
    @property
    def t(self):
        return self.toks[self.i]

    def peek(self, n=1):
        return self.toks[min(self.i + n, len(self.toks) - 1)]

    def err(self, msg):
        raise SyntaxError("%s:%d: %s (at %r)" % (self.fname, self.t.line, msg, self.t.val))

    def is_op(self, v):
        return self.t.kind == "op" and self.t.val == v

    def is_kw(self, v):
        return self.t.kind == "kw" and self.t.val == v

    def accept(self, v):
        if self.t.kind in ("op", "kw") and self.t.val == v:
            self.i += 1
            return True
        return False

    def expect(self, v):
        if not self.accept(v):
            self.err("expected %r" % v)

    def ident(self):
        if self.t.kind != "id":
            self.err("expected identifier")
        v = self.t.val
        self.i += 1
        return v

    def skip_semis(self):
        while self.is_op(";"):
            self.i += 1

    # ---- file ----
    def parse_file(self):
        self.skip_semis()
        self.expect("package")
        pkg = self.ident()
        self.skip_semis()
        imports = []
        while self.is_kw("import"):
            self.i += 1
            if self.accept("("):
                while not self.is_op(")"):
                    imports.append(self.import_spec())
                    self.skip_semis()
                self.expect(")")
            else:
                imports.append(self.import_spec())
            self.skip_semis()
        decls = []
        while self.t.kind != "eof":
            decls.extend(self.top_decl())
            self.skip_semis()
        return ("file", pkg, imports, decls)

    def import_spec(self):
        name = None
        if self.t.kind == "id" or self.is_op("."):
            name = self.t.val
            self.i += 1
        path = self.t.val[1:-1]
        self.i += 1
        return (name or path.split("/")[-1], path)

    def top_decl(self):
        if self.is_kw("func"):
            return [self.func_decl()]
        return self.gen_decl()

    def gen_decl(self):
        kw = self.t.val
        line = self.t.line
        self.i += 1
        out = []
        if self.accept("("):
            idx = 0
            prev = None
            while not self.is_op(")"):
                d = self.spec(kw, idx, prev, line)
                if kw == "const":
                    prev = d
                out.append(d)
                idx += 1
                self.skip_semis()
            self.expect(")")
        else:
            out.append(self.spec(kw, 0, None, line))
        return out

    def spec(self, kw, idx, prev, line):
        line = self.t.line
        if kw == "type":
            name = self.ident()
            tparams = None
            if self.is_op("[") and self.peek().kind == "id" and not (self.peek(2).kind == "op" and self.peek(2).val == "]"):
                # generic type: not used by the translated sources
                self.err("generic type declarations are not supported")
            alias = self.accept("=")
            ty = self.type_()
            return ("typedecl", name, ty, alias, line)
        names = [self.ident()]
        while self.accept(","):
            names.append(self.ident())
        ty = None
        vals = None
        if not self.is_op("=") and not self.is_op(";") and not self.is_op(")"):
            ty = self.type_()
        if self.accept("="):
            vals = self.expr_list()
        if kw == "const":
            implicit = False
            if vals is None:  # implicit repetition of the previous spec's expression list and type
                if prev is None:
                    self.err("const without value")
                ty, vals, implicit = prev[2], prev[3], True
            return ("const", names, ty, vals, idx, line, implicit)
        return ("var", names, ty, vals, line)

    def func_decl(self):
        line = self.t.line
        self.expect("func")
        recv = None
        if self.is_op("("):
            self.i += 1
            rname = None
            if self.t.kind == "id" and not (self.peek().kind == "op" and self.peek().val in (")", ".")):
                rname = self.ident()
            rty = self.type_()
            self.expect(")")
            recv = (rname, rty)
        name = self.ident()
        tparams = None
        if self.is_op("["):  # type parameters: [I Indexer, ...]
            self.i += 1
            tparams = []
            while not self.is_op("]"):
                ns = [self.ident()]
                while self.accept(","):
                    ns.append(self.ident())
                cons = self.type_()
                tparams.extend((n, cons) for n in ns)
                self.accept(",")
            self.expect("]")
        sig = self.signature()
        body = None
        if self.is_op("{"):
            body = self.block()
        return ("func", name, recv, sig, body, tparams, line)

    def signature(self):
        params = self.params()
        results = []
        if self.is_op("("):
            results = self.params()
        elif not (self.is_op("{") or self.is_op(";") or self.is_op(")") or self.is_op(",") or self.is_op("]") or self.is_op("}")
                  or self.is_op("=") or self.is_op(":=") or self.t.kind == "str" or self.t.kind == "eof"):
            results = [(None, self.type_(), False)]
        return ("sig", params, results)

    def params(self):
        """(name, type, variadic) list.  Go groups `a, b int`; unnamed lists hold types only."""
        self.expect("(")
        items = []
        while not self.is_op(")"):
            variadic = self.accept("...")
            ty = self.type_()
            name = None
            if not variadic and not self.is_op(",") and not self.is_op(")"):
                # `ty` was a name, now the type follows
                if ty[0] != "name" or ty[1] is not None:
                    self.err("bad parameter")
                name = ty[2]
                variadic = self.accept("...")
                ty = self.type_()
            items.append([name, ty, variadic])
            if not self.accept(","):
                break
        self.expect(")")
        # resolve grouping: if any item is named, unnamed items before it are names sharing the next type
        if any(it[0] is not None for it in items):
            out = []
            pending = []
            for name, ty, variadic in items:
                if name is None:
                    if ty[0] != "name" or ty[1] is not None:
                        self.err("mixed named and unnamed parameters")
                    pending.append(ty[2])
                else:
                    for p in pending:
                        out.append((p, ty, variadic))
                    pending = []
                    out.append((name, ty, variadic))
            if pending:
                self.err("mixed named and unnamed parameters")
            return out
        return [tuple(it) for it in items]

    # ---- types ----
    def type_(self):
        t = self.t
        if t.kind == "id":
            name = self.ident()
            if self.is_op(".") and self.peek().kind == "id":
                self.i += 1
                return ("name", name, self.ident())
            return ("name", None, name)
        if self.accept("*"):
            return ("ptr", self.type_())
        if self.accept("("):
            ty = self.type_()
            self.expect(")")
            return ty
        if self.is_op("["):
            self.i += 1
            if self.accept("]"):
                return ("slice", self.type_())
            if self.accept("..."):
                self.expect("]")
                return ("array", None, self.type_())
            self.exprlev += 1
            n = self.expr()
            self.exprlev -= 1
            self.expect("]")
            return ("array", n, self.type_())
        if self.accept("map"):
            self.expect("[")
            k = self.type_()
            self.expect("]")
            return ("map", k, self.type_())
        if self.accept("chan"):
            self.accept("<-")
            return ("chan", self.type_())
        if self.is_op("<-"):
            self.i += 1
            self.expect("chan")
            return ("chan", self.type_())
        if self.accept("func"):
            return ("functype", self.signature())
        if self.accept("struct"):
            return self.struct_type()
        if self.accept("interface"):
            return self.interface_type()
        self.err("expected type")

    def struct_type(self):
        self.expect("{")
        fields = []  # (name or None for embedded, type)
        self.skip_semis()
        while not self.is_op("}"):
            if self.is_op("*") or (self.t.kind == "id" and self.peek().kind == "op" and self.peek().val in (";", "}", ".")) or \
               (self.t.kind == "id" and self.peek().kind == "str"):
                ty = self.type_()
                fields.append((None, ty))
            else:
                names = [self.ident()]
                while self.accept(","):
                    names.append(self.ident())
                ty = self.type_()
                for n in names:
                    fields.append((n, ty))
            if self.t.kind == "str":  # tag
                self.i += 1
            self.skip_semis()
        self.expect("}")
        return ("struct", fields)

    def interface_type(self):
        self.expect("{")
        methods = []
        embeds = []
        self.skip_semis()
        while not self.is_op("}"):
            if self.t.kind == "id" and self.peek().kind == "op" and self.peek().val == "(":
                name = self.ident()
                methods.append((name, self.signature()))
            else:  # embedded interface or a type-set constraint (a | b | ~c)
                self.accept("~")
                ty = self.type_()
                while self.accept("|"):
                    self.accept("~")
                    self.type_()
                embeds.append(ty)
            self.skip_semis()
        self.expect("}")
        return ("interface", methods, embeds)

    # ---- statements ----
    def block(self):
        self.expect("{")
        saved = self.exprlev
        self.exprlev = 0
        stmts = self.stmt_list()
        self.exprlev = saved
        self.expect("}")
        return ("block", stmts)

    def stmt_list(self):
        out = []
        self.skip_semis()
        while not self.is_op("}") and not self.is_kw("case") and not self.is_kw("default") and self.t.kind != "eof":
            s = self.stmt()
            if s is not None:
                out.append(s)
            self.skip_semis()
        return out

    def stmt(self):
        t = self.t
        line = t.line
        if t.kind == "kw":
            v = t.val
            if v in ("var", "const", "type"):
                return ("declstmt", self.gen_decl(), line)
            if v == "return":
                self.i += 1
                vals = []
                if not self.is_op(";") and not self.is_op("}"):
                    vals = self.expr_list()
                return ("return", vals, line)
            if v == "if":
                return self.if_stmt()
            if v == "for":
                return self.for_stmt(None)
            if v == "switch":
                return self.switch_stmt(None)
            if v in ("break", "continue", "goto"):
                self.i += 1
                label = None
                if self.t.kind == "id":
                    label = self.ident()
                return (v, label, line)
            if v == "fallthrough":
                self.i += 1
                return ("fallthrough", line)
            if v == "defer":
                self.i += 1
                return ("defer", self.expr(), line)
            if v == "go":
                self.i += 1
                return ("go", self.expr(), line)
            if v == "func":
                return self.simple_stmt()
            if v == "select":
                # goroutine plumbing: parsed as an opaque statement (its tokens are skipped to the matching brace) so that a file can
                # be read for its other declarations; go2cpp refuses a function that contains one
                self.i += 1
                depth = 0
                while True:
                    if self.is_op("{"):
                        depth += 1
                    elif self.is_op("}"):
                        depth -= 1
                        if depth == 0:
                            self.i += 1
                            break
                    self.i += 1
                return ("select", line)
            self.err("unexpected keyword")
        if self.is_op("{"):
            return self.block()
        if self.is_op(";"):
            return None
        if t.kind == "id" and self.peek().kind == "op" and self.peek().val == ":" and not (self.peek(2).kind == "op" and self.peek(2).val == "="):
            label = self.ident()
            self.expect(":")
            self.skip_semis()
            if self.is_kw("for"):
                return self.for_stmt(label)
            if self.is_kw("switch"):
                return self.switch_stmt(label)
            if self.is_op("}"):
                return ("labeled", label, None, line)
            return ("labeled", label, self.stmt(), line)
        return self.simple_stmt()

    def simple_stmt(self, range_ok=False):
        line = self.t.line
        if range_ok and self.is_kw("range"):
            self.i += 1
            return ("range", [], False, self.expr(), line)
        lhs = self.expr_list()
        t = self.t
        if t.kind == "op":
            v = t.val
            if v in (":=", "="):
                self.i += 1
                if range_ok and self.is_kw("range"):
                    self.i += 1
                    return ("range", lhs, v == ":=", self.expr(), line)
                rhs = self.expr_list()
                return ("define" if v == ":=" else "assign", lhs, rhs, line)
            if v in ("+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>=", "&^="):
                self.i += 1
                rhs = self.expr()
                return ("opassign", v[:-1], lhs[0], rhs, line)
            if v in ("++", "--"):
                self.i += 1
                return ("incdec", v, lhs[0], line)
            if v == "<-":
                self.i += 1
                return ("send", lhs[0], self.expr(), line)
        if len(lhs) != 1:
            self.err("expression list used as statement")
        return ("exprstmt", lhs[0], line)

    def header(self):
        """[init;] cond of if / switch; composite literals of bare names are off."""
        saved = self.exprlev
        self.exprlev = -1
        init = None
        cond = None
        if not self.is_op("{"):
            if self.is_op(";"):
                pass
            else:
                init = self.simple_stmt()
            if self.accept(";"):
                if not self.is_op("{"):
                    cond = self.simple_stmt()
            else:
                cond, init = init, None
        self.exprlev = saved
        return init, cond

    def if_stmt(self):
        line = self.t.line
        self.expect("if")
        init, cond = self.header()
        if cond is None or cond[0] != "exprstmt":
            self.err("missing condition in if")
        body = self.block()
        els = None
        if self.accept("else"):
            els = self.if_stmt() if self.is_kw("if") else self.block()
        return ("if", init, cond[1], body, els, line)

    def for_stmt(self, label):
        line = self.t.line
        self.expect("for")
        saved = self.exprlev
        self.exprlev = -1
        init = cond = post = None
        rng = None
        if not self.is_op("{"):
            if not self.is_op(";"):
                s = self.simple_stmt(range_ok=True)
                if s[0] == "range":
                    rng = s
                else:
                    init = s
            if rng is None:
                if self.accept(";"):
                    if not self.is_op(";"):
                        c = self.simple_stmt()
                        cond = c[1]
                    self.expect(";")
                    if not self.is_op("{"):
                        post = self.simple_stmt()
                else:
                    cond, init = init[1], None
        self.exprlev = saved
        body = self.block()
        if rng is not None:
            return ("forrange", label, rng[1], rng[2], rng[3], body, line)
        return ("for", label, init, cond, post, body, line)

    def switch_stmt(self, label):
        line = self.t.line
        self.expect("switch")
        init, tag = self.header()
        tagexpr = None
        if tag is not None:
            if tag[0] == "exprstmt":
                tagexpr = tag[1]
            else:
                self.err("type switches are not supported")
        self.expect("{")
        cases = []
        self.skip_semis()
        while not self.is_op("}"):
            if self.accept("default"):
                exprs = None
            else:
                self.expect("case")
                exprs = self.expr_list()
            self.expect(":")
            body = self.stmt_list()
            cases.append((exprs, body))
        self.expect("}")
        return ("switch", label, init, tagexpr, cases, line)

    # ---- expressions ----
    def expr_list(self):
        out = [self.expr()]
        while self.accept(","):
            out.append(self.expr())
        return out

    def expr(self, prec=1):
        x = self.unary()
        while True:
            t = self.t
            if t.kind != "op" or t.val not in BINPREC or BINPREC[t.val] < prec:
                return x
            op = t.val
            self.i += 1
            y = self.expr(BINPREC[op] + 1)
            x = ("binary", op, x, y)

    def unary(self):
        t = self.t
        if t.kind == "op" and t.val in ("+", "-", "!", "^", "*", "&", "<-"):
            self.i += 1
            x = self.unary()
            return ("unary", t.val, x)
        return self.primary()

    def operand(self):
        t = self.t
        if t.kind == "int":
            self.i += 1
            return ("int", t.val)
        if t.kind == "float":
            self.i += 1
            return ("float", t.val)
        if t.kind == "rune":
            self.i += 1
            return ("rune", t.val)
        if t.kind == "str":
            self.i += 1
            return ("str", t.val)
        if t.kind == "id":
            self.i += 1
            return ("ident", t.val)
        if t.kind == "op" and t.val == "(":
            self.i += 1
            saved = self.exprlev
            self.exprlev = 1
            # a parenthesised type, e.g. (*T)(x)
            if self.is_op("["):  # ((*T)(x) is told from (*p) by the translator, which knows the type names)
                save_i = self.i
                try:
                    ty = self.type_()
                    if self.is_op(")"):
                        self.i += 1
                        self.exprlev = saved
                        return ("typeexpr", ty)
                except SyntaxError:
                    pass
                self.i = save_i
            x = self.expr()
            self.exprlev = saved
            self.expect(")")
            return ("paren", x)
        if t.kind == "kw" and t.val == "func":
            self.i += 1
            sig = self.signature()
            if self.is_op("{"):
                saved = self.exprlev
                self.exprlev = 0
                body = self.block()
                self.exprlev = saved
                return ("funclit", sig, body)
            return ("typeexpr", ("functype", sig))
        if (t.kind == "op" and t.val == "[") or (t.kind == "kw" and t.val in ("struct", "map", "chan", "interface")):
            ty = self.type_()
            return ("typeexpr", ty)
        self.err("unexpected token in expression")

    def primary(self):
        x = self.operand()
        while True:
            t = self.t
            if t.kind != "op":
                return x
            v = t.val
            if v == ".":
                self.i += 1
                if self.accept("("):
                    if self.is_kw("type"):
                        self.err("type switch")
                    ty = self.type_()
                    self.expect(")")
                    x = ("typeassert", x, ty)
                else:
                    x = ("selector", x, self.ident())
            elif v == "[":
                self.i += 1
                saved = self.exprlev
                self.exprlev = 1
                lo = hi = mx = None
                if not self.is_op(":"):
                    lo = self.expr()
                if self.accept(":"):
                    if not self.is_op("]") and not self.is_op(":"):
                        hi = self.expr()
                    three = False
                    if self.accept(":"):
                        three = True
                        mx = self.expr()
                    self.exprlev = saved
                    self.expect("]")
                    x = ("slice", x, lo, hi, mx, three)
                else:
                    self.exprlev = saved
                    self.expect("]")
                    x = ("index", x, lo)
            elif v == "(":
                self.i += 1
                saved = self.exprlev
                self.exprlev = 1
                args = []
                ell = False
                while not self.is_op(")"):
                    # a type as argument (make([]T, n), new(T))
                    if self.is_op("*") and x == ("ident", "new"):
                        args.append(("typeexpr", self.type_()))
                    else:
                        args.append(self.expr())
                    if self.accept("..."):
                        ell = True
                    if not self.accept(","):
                        break
                self.exprlev = saved
                self.expect(")")
                x = ("call", x, args, ell)
            elif v == "{":
                if not self.lit_type_ok(x):
                    return x
                x = ("complit", self.as_type(x), self.lit_value())
            else:
                return x

    def lit_type_ok(self, x):
        if x[0] == "typeexpr":
            return x[1][0] in ("array", "slice", "map", "struct", "name")
        if self.exprlev < 0:
            return False
        if x[0] == "ident":
            return True
        if x[0] == "selector" and x[1][0] == "ident":
            return True
        return False

    def as_type(self, x):
        if x[0] == "typeexpr":
            return x[1]
        if x[0] == "ident":
            return ("name", None, x[1])
        if x[0] == "selector":
            return ("name", x[1][1], x[2])
        self.err("not a type")

    def lit_value(self):
        self.expect("{")
        saved = self.exprlev
        self.exprlev = 1
        elems = []  # (key or None, value) — value may itself be ("litval", elems) for elided types
        self.skip_semis()
        while not self.is_op("}"):
            k = None
            v = self.lit_elem()
            if self.accept(":"):
                k = v
                v = self.lit_elem()
            elems.append((k, v))
            if not self.accept(","):
                self.skip_semis()
                break
            self.skip_semis()
        self.exprlev = saved
        self.expect("}")
        return elems

    def lit_elem(self):
        if self.is_op("{"):
            return ("litval", self.lit_value())
        return self.expr()


def parse(src, fname="?"):
    return Parser(src, fname).parse_file()


if __name__ == "__main__":
    import sys
    for f in sys.argv[1:]:
        ast = parse(open(f).read(), f)
        print(f, "ok:", len(ast[3]), "declarations")
