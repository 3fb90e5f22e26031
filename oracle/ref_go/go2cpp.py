"""go2cpp.py — syntax-directed translation of the reference's pure-Go encoder sources into C++ (build time, test infrastructure).

    python oracle/ref_go/go2cpp.py <reference root> <output .cpp>

What the reference's zstd encoder computes is defined by Go source only (no assembly on the encode side except matchLen / XXH64,
which have pure-Go twins).  No Go toolchain exists in the build image, so — like ref_s2asm/plan9_to_gas.py re-spells the S2
assembly for the GNU assembler — this script re-spells the Go statements as C++ statements, one for one, at build time; gort.h
supplies Go's value semantics (typed wrap-around integers, untyped constants, slices, arrays, interfaces) so that the C++
compiler does the type checking Go would do.  Nothing is interpreted, reordered or optimised; the output goes under oracle/_ref/
(git-ignored) and the only hand-written code beside it is driver.cpp (option structs -> one encodeAll call).

The translation is purely syntactic: no Go type checker.  Where C++ needs to know more than the syntax says, the rule is uniform:
  * every selector a.b is written a->b (structs, named integers and slices define operator-> returning themselves, so values
    and pointers read alike); a.b with `a` an imported package is pkg::b;
  * every integer literal is an untyped constant go::K(...); `x := e` is `auto x = go::def(e)` (untyped constants become int);
  * every binary expression is fully parenthesised (Go's operator precedence differs from C++'s);
  * switch statements become if / else-if chains (no fallthrough in these sources), `break` inside them a goto;
  * loops become `for (;;)` with the condition, body, continue label and post statement spelled out, so that labelled
    break / continue are plain gotos;
  * methods are collected per receiver type (Go declares them anywhere in the package) and become member functions; embedded
    structs become base classes (promotion of fields and methods == inheritance, shadowing == hiding);
  * an interface type becomes a type-erased handle with one adapter template (calls resolve on the concrete type exactly like
    Go's method sets).
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import goparse  # noqa: E402

BUILTIN_TYPES = {"int": "Int", "uint": "Uint", "int8": "int8", "int16": "int16", "int32": "int32", "int64": "int64", "uint8": "uint8",
                 "uint16": "uint16", "uint32": "uint32", "uint64": "uint64", "uintptr": "uintptr", "byte": "byte", "rune": "rune",
                 "bool": "bool", "string": "String", "float64": "float64", "float32": "float32", "error": "error", "any": "go::Any"}
CPP_RESERVED = {"new", "delete", "this", "class", "template", "typename", "union", "register", "int", "bool", "char", "short", "long",
                "float", "double", "signed", "unsigned", "void", "auto", "operator", "private", "public", "protected", "friend",
                "namespace", "using", "virtual", "inline", "static", "extern", "volatile", "do", "while", "try", "catch", "throw",
                "enum", "typedef", "sizeof", "and", "or", "not", "xor", "export", "mutable", "explicit", "asm", "near", "far",
                "errno", "stdin", "stdout", "stderr", "NULL", "EOF", "assert", "min", "max", "len", "cap", "copy", "append", "panic",
                "nil", "I", "K", "Slice", "Array", "String", "Int", "Uint", "uint8", "uint16", "uint32", "uint64", "int8", "int16",
                "int32", "int64", "byte", "rune", "error", "idx", "def", "main", "signal", "index", "abs", "log", "exp", "floor", "ceil", "round",
                "time", "clock", "rand", "random", "exit", "abort", "free", "malloc", "calloc", "div", "remove", "rename", "link", "read", "write"}
BUILTIN_FUNCS = {"len", "cap", "append", "copy", "make", "new", "panic", "min", "max", "print", "println", "delete", "close", "clear", "recover"}
LIBRARY_PKGS = {"crc32", "bits", "binary", "math", "fmt", "errors", "bytes", "sync", "io", "rand", "race", "le", "hex", "strings", "strconv", "os", "runtime", "sort", "cpuinfo", "debug", "unsafe", "hash", "bufio", "log", "time", "atomic"}


class Unsupported(Exception):
    pass


def go_int(lit):
    s = lit.replace("_", "")
    if s.startswith(("0x", "0X", "0b", "0B", "0o", "0O")):
        return int(s, 0)
    if len(s) > 1 and s[0] == "0":
        return int(s, 8)
    return int(s, 10)


def go_rune(lit):
    body = lit[1:-1]
    if body[0] != "\\":
        return ord(body)
    esc = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11, "\\": 92, "'": 39, '"': 34}
    c = body[1]
    if c in esc:
        return esc[c]
    if c == "x":
        return int(body[2:4], 16)
    if c in "uU":
        return int(body[2:], 16)
    return int(body[1:4], 8)


def c_string(lit):
    if lit[0] == "`":
        body = lit[1:-1]
        out = body.replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n").replace("\r", "")
        return '"' + out + '"'
    # Go's escapes \a \b \f \n \r \t \v \\ \" \xNN \NNN are C's; \uNNNN: keep as is (UTF-8 source charset)
    return lit


def mangle(n):
    if not n.isascii():
        n = "".join(c if c.isascii() else "u%04x" % ord(c) for c in n)
    if n in CPP_RESERVED:
        return n + "_"
    return n


class Package:
    def __init__(self, name, files):
        self.name = name          # Go package name
        self.files = files        # [(path, ast)]
        self.types = {}           # name -> typedecl node
        self.funcs = {}           # name -> func node (no receiver)
        self.methods = {}         # receiver type name -> [func node]
        self.consts = []          # const nodes in order
        self.vars = []            # var nodes in order
        self.inits = []           # init functions
        self.method_names = set()
        self.field_names = set()
        self.var_names = set()
        self.bool_consts = {}
        self.embedded_names = set()
        self.local_names = set()


class Emitter:
    def __init__(self, pkgs, cfg):
        self.pkgs = {p.name: p for p in pkgs}
        self.cfg = cfg
        self.out = []
        self.ind = 0
        self.tmp = 0
        self.scopes = []
        self.pkg = None
        self.imports = {}
        self.ctx = []         # stack of ("loop", continue_label) / ("switch", end_label)
        self.labels = {}
        self.named_results = None
        self.cur_results = None
        self.warnings = []
        self.local_types = []
        self.heap_vars = set()
        self.sliced_vars = set()
        self.goto_targets = set()

    # ---- output helpers ----
    def w(self, s=""):
        self.out.append("    " * self.ind + s if s else "")

    def newtmp(self, p="t"):
        self.tmp += 1
        return "__%s%d" % (p, self.tmp)

    def push(self):
        self.scopes.append(set())

    def pop(self):
        self.scopes.pop()

    def declare(self, n):
        if self.scopes:
            self.scopes[-1].add(n)

    def is_local(self, n):
        return any(n in s for s in self.scopes)

    # ---- types ----
    def is_type_name(self, n):
        if self.is_local(n):
            return False
        return n in BUILTIN_TYPES or n in self.pkg.types

    def tname(self, pkg, n):
        """C++ name of a declared Go type: Go keeps types and fields / methods / variables in separate positions, C++ does not
        (`cTable cTable;` in a struct changes the meaning of the name), so a type whose name is also a member or variable name in
        its package gets a suffix."""
        if pkg is not None and (n in pkg.field_names or n in pkg.method_names or n in pkg.funcs or n in pkg.var_names or n in pkg.local_names):
            return mangle(n) + "_T"
        return mangle(n)

    def ctype(self, t):
        k = t[0]
        if k == "name":
            if t[1] is None:
                n = t[2]
                if n in BUILTIN_TYPES and n not in self.pkg.types:
                    return BUILTIN_TYPES[n]
                if n in self.pkg.types:
                    return self.tname(self.pkg, n)
                return mangle(n)
            p = self.pkgs.get(self.imports.get(t[1]))
            return "%s::%s" % (mangle(t[1]), self.tname(p, t[2]) if p is not None and t[2] in p.types else mangle(t[2]))
        if k == "ptr":
            return self.ctype(t[1]) + "*"
        if k == "slice":
            return "Slice<%s>" % self.ctype(t[1])
        if k == "array":
            if t[1] is None:
                raise Unsupported("[...]T outside a composite literal")
            return "Array<%s, go::cn(%s)>" % (self.ctype(t[2]), self.ex(t[1]))
        if k == "functype":
            sig = t[1]
            return "std::function<%s(%s)>" % (self.result_type(sig[2]), ", ".join(self.param_type(p) for p in sig[1]))
        if k == "struct":
            return self.anon_struct(t)
        if k == "interface":
            if not t[1] and not t[2]:
                return "go::Any"
            raise Unsupported("anonymous interface type")
        if k == "map":
            return "go::Map<%s, %s>" % (self.ctype(t[1]), self.ctype(t[2]))
        if k == "chan":
            return "go::Chan<%s>" % self.ctype(t[1])
        raise Unsupported("type %r" % (k,))

    def anon_struct(self, t):
        fields = "; ".join("%s %s{}" % (self.ctype(ft), mangle(fn)) for fn, ft in t[1])
        return "struct { %s; auto operator->() { return this; } }" % fields

    def param_type(self, p):
        name, ty, variadic = p
        if variadic:
            return "Slice<%s>" % self.ctype(ty)
        return self.ctype(ty)

    def result_type(self, results):
        if not results:
            return "void"
        if len(results) == 1:
            return self.ctype(results[0][1])
        return "std::tuple<%s>" % ", ".join(self.ctype(r[1]) for r in results)

    # ---- expressions ----
    def ex(self, e):
        k = e[0]
        m = getattr(self, "ex_" + k, None)
        if m is None:
            raise Unsupported("expression %r" % (k,))
        return m(e)

    def ex_int(self, e):
        v = go_int(e[1])
        if v > 0x7FFFFFFFFFFFFFFF:
            return "K(%dULL)" % v
        return "K(%dLL)" % v

    def ex_rune(self, e):
        return "K(%dLL)" % go_rune(e[1])

    def ex_float(self, e):
        s = e[1].replace("_", "")
        if s.startswith("."):
            s = "0" + s
        if "." not in s and "e" not in s.lower():
            s += ".0"
        return s

    def ex_str(self, e):
        lit = c_string(e[1])
        if re.search(r"\\(x00|0(?![0-7])|00(?![0-7])|000)", lit):  # a NUL inside the literal: a C string would end there (s2's magic chunk)
            return "String(std::string(%s, sizeof(%s) - 1))" % (lit, lit)
        return "String(%s)" % lit

    def ex_ident(self, e):
        n = e[1]
        if not self.is_local(n):
            if n == "nil":
                return "go::nil"
            if n in ("true", "false"):
                return n
            if n == "iota":
                return "K(%dLL)" % self.iota
            if n in BUILTIN_TYPES and n not in self.pkg.types:
                return BUILTIN_TYPES[n]
            if n in getattr(self, "recv_methods", ()) and n in self.pkg.funcs:
                return "%s::%s" % (mangle(self.pkg.name), mangle(n))
        return mangle(n)

    def ex_paren(self, e):
        return "(%s)" % self.ex(e[1])

    def ex_typeexpr(self, e):
        return self.ctype(e[1])

    def ex_binary(self, e):
        op, a, b = e[1], self.ex(e[2]), self.ex(e[3])
        if op == "&^":
            return "(%s & ~(%s))" % (a, b)
        return "(%s %s %s)" % (a, op, b)

    def ex_unary(self, e):
        op, x = e[1], e[2]
        if op == "&":
            if x[0] == "complit":
                return self.complit(x, heap=True)
            if x[0] == "paren" and x[1][0] == "complit":
                return self.complit(x[1], heap=True)
            return "(&%s)" % self.ex(x)
        if op == "^":
            return "(~%s)" % self.ex(x)
        if op == "<-":
            return "go::recv(%s)" % self.ex(x)
        return "(%s%s)" % (op, self.ex(x))

    def is_pkg(self, x):
        return x[0] == "ident" and x[1] in self.imports and not self.is_local(x[1])

    def ex_selector(self, e, call=False):
        x, name = e[1], e[2]
        if self.is_pkg(x):
            return "%s::%s" % (mangle(x[1]), mangle(name))
        base = self.ex(x)
        if name in self.pkg.embedded_names and name not in self.pkg.field_names and name in self.pkg.types:
            return "go::base<%s>(%s)" % (self.tname(self.pkg, name), base)  # e.fastEncoder: the embedded struct == the base class
        if not call and name in self.all_method_names and name not in self.all_field_names:
            # method value: a closure over the receiver
            return "[=](auto&&... __a) { return %s->%s(std::forward<decltype(__a)>(__a)...); }" % (base, mangle(name))
        return "%s->%s" % (base, mangle(name))

    def ex_index(self, e):
        return "go::ix(%s)[%s]" % (self.ex(e[1]), self.ex(e[2]))  # (ix: Go indexes through a pointer to an array)

    def ex_slice(self, e):
        x = self.ex(e[1])
        lo = "go::idx(%s)" % self.ex(e[2]) if e[2] is not None else "-1"
        hi = "go::idx(%s)" % self.ex(e[3]) if e[3] is not None else "-1"
        if e[5]:
            return "go::slice3(%s, %s, %s, go::idx(%s))" % (x, lo, hi, self.ex(e[4]))
        return "go::slice(%s, %s, %s)" % (x, lo, hi)

    def callee_info(self, fun):
        """(variadic position or None) of a user-defined callee known by name."""
        name = None
        if fun[0] == "ident" and not self.is_local(fun[1]):
            f = self.pkg.funcs.get(fun[1])
            if f is not None:
                name = f
        elif fun[0] == "selector":
            if self.is_pkg(fun[1]):
                p = self.pkgs.get(self.imports[fun[1][1]])
                if p is not None:
                    name = p.funcs.get(fun[2])
            else:
                cands = self.all_methods.get(fun[2], [])
                if cands:
                    name = cands[0]
        if name is None:
            return None
        params = name[3][1]
        for i, p in enumerate(params):
            if p[2]:
                return i, p[1]
        return None

    def ex_call(self, e):
        fun, args, ell = e[1], e[2], e[3]
        # builtins
        if fun[0] == "ident" and fun[1] in BUILTIN_FUNCS and not self.is_local(fun[1]) and fun[1] not in self.pkg.funcs:
            return self.builtin_call(fun[1], args, ell)
        # conversions
        ty = None
        if fun[0] == "typeexpr":
            ty = fun[1]
        elif fun[0] == "paren" and fun[1][0] == "typeexpr":
            ty = fun[1][1]
        elif fun[0] == "paren" and fun[1][0] == "unary" and fun[1][1] == "*":  # (*T)(x)
            inner = fun[1][2]
            if inner[0] == "ident" and self.is_type_name(inner[1]):
                ty = ("ptr", ("name", None, inner[1]))
            elif inner[0] == "typeexpr":
                ty = ("ptr", inner[1])
        elif fun[0] == "ident" and self.is_type_name(fun[1]):
            ty = ("name", None, fun[1])
        elif fun[0] == "selector" and self.is_pkg(fun[1]) and self.is_external_type(fun[1][1], fun[2]):
            ty = ("name", fun[1][1], fun[2])
        if ty is not None:
            return self.conversion(ty, args[0])
        if fun[0] == "selector":
            f = self.ex_selector(fun, call=True)
        else:
            f = self.ex(fun)
        cargs = [self.ex(a) for a in args]
        vi = self.callee_info(fun)
        if vi is not None and not ell:
            pos, ety = vi
            packed = "Slice<%s>{%s}" % (self.ctype(ety), ", ".join("%s(%s)" % (self.ctype(ety), a) if False else a for a in cargs[pos:]))
            if len(cargs) <= pos:
                packed = "Slice<%s>()" % self.ctype(ety)
            cargs = cargs[:pos] + [packed]
        return "%s(%s)" % (f, ", ".join(cargs))

    def is_external_type(self, pkgname, name):
        p = self.pkgs.get(self.imports.get(pkgname))
        if p is not None:
            return name in p.types
        return (pkgname, name) in {("io", "Writer"), ("io", "Reader"), ("sync", "Once"), ("sync", "Mutex"), ("sync", "WaitGroup"), ("sync", "Pool")}

    def conversion(self, ty, arg):
        a = self.ex(arg)
        if ty[0] == "slice" and ty[1] == ("name", None, "byte"):
            return "go::to_bytes(%s)" % a
        if ty == ("name", None, "string") and "string" not in self.pkg.types:
            return "go::to_string(%s)" % a
        if ty == ("name", None, "float64") or ty == ("name", None, "float32"):
            return "(%s)go::f64(%s)" % (BUILTIN_TYPES[ty[2]], a)
        if ty[0] == "ptr" and self.underlying(ty[1])[0] == "array":  # (*[N]T)(slice): Go 1.17 slice-to-array-pointer conversion
            return "go::as_array<%s>(%s)" % (self.ctype(ty[1]), a)
        if ty[0] == "ptr":
            return "((%s)(%s))" % (self.ctype(ty), a)
        return "%s(%s)" % (self.ctype(ty), a)

    def builtin_call(self, name, args, ell):
        if name == "len":
            return "go::len(%s)" % self.ex(args[0])
        if name == "cap":
            return "go::cap(%s)" % self.ex(args[0])
        if name == "append":
            if ell:
                return "go::append_all(%s, %s)" % (self.ex(args[0]), self.ex(args[1]))
            return "go::append(%s)" % ", ".join(self.ex(a) for a in args)
        if name == "copy":
            return "go::copy(%s, %s)" % (self.ex(args[0]), self.ex(args[1]))
        if name == "make":
            a0 = args[0]
            ty = a0[1] if a0[0] == "typeexpr" else (("name", None, a0[1]) if a0[0] == "ident" else (("name", a0[1][1], a0[2]) if a0[0] == "selector" else None))
            u = self.underlying(ty) if ty is not None else None
            if u is None or u[0] != "slice":
                raise Unsupported("make of a non-slice")
            n = "go::idx(%s)" % self.ex(args[1])
            c = ", go::idx(%s)" % self.ex(args[2]) if len(args) > 2 else ""
            made = "go::make_slice<%s>(%s%s)" % (self.ctype(u[1]), n, c)
            return made if ty[0] == "slice" else "%s(%s)" % (self.ctype(ty), made)
        if name == "new":
            a = args[0]
            ty = a[1] if a[0] == "typeexpr" else (("name", None, a[1]) if a[0] == "ident" else ("name", a[1][1], a[2]))
            return "(new (go::rt::tag) %s())" % self.ctype(ty)
        if name == "panic":
            return "go::panic(%s)" % self.ex(args[0])
        if name in ("min", "max"):
            r = self.ex(args[0])
            for a in args[1:]:
                r = "go::%s(%s, %s)" % (name, r, self.ex(a))
            return r
        if name == "clear":
            return "go::clear(%s)" % self.ex(args[0])
        if name == "recover":
            return "go::recover_()"  # a panic of the translated code is a C++ exception that travels to the driver: nothing to recover
        if name in ("print", "println"):
            return "go::println(%s)" % ", ".join(self.ex(a) for a in args)
        raise Unsupported("builtin %s" % name)

    def ex_funclit(self, e, capture="&"):
        sig, body = e[1], e[2]
        params = ", ".join("%s %s" % (self.param_type(p), mangle(p[0]) if p[0] and p[0] != "_" else self.newtmp("u")) for p in sig[1])
        saved = (self.out, self.ind, self.ctx, self.named_results, self.cur_results, self.labels)
        self.out, self.ind, self.ctx, self.labels = [], self.ind + 1, [], {}
        self.push()
        for p in sig[1]:
            if p[0]:
                self.declare(p[0])
        self.func_prologue(sig)
        self.block_body(body)
        self.pop()
        text = "\n".join(self.out)
        self.out, self.ind, self.ctx, self.named_results, self.cur_results, self.labels = saved
        return "[%s](%s) -> %s {\n%s\n%s}" % (capture, params, self.result_type(sig[2]), text, "    " * self.ind)

    def ex_complit(self, e):
        return self.complit(e, heap=False)

    def ex_typeassert(self, e):
        return "go::cast<%s>(%s)" % (self.ctype(e[2]), self.ex(e[1]))

    def struct_fields(self, tname_node):
        """Field names in declaration order (embedded types by their type name) of a named struct type, or None."""
        if tname_node[0] != "name":
            return None
        p = self.pkg if tname_node[1] is None else self.pkgs.get(self.imports.get(tname_node[1]))
        if p is None:
            return None
        td = p.types.get(tname_node[2])
        if td is None:
            return None
        ty = td[2]
        if ty[0] != "struct":
            return None
        out = []
        for fn, ft in ty[1]:
            if fn is None:
                base = ft[1] if ft[0] == "ptr" else ft
                out.append(base[2])
            else:
                out.append(fn)
        return out

    def underlying(self, t):
        """Resolve a named type to its declared underlying type expression (within known packages)."""
        seen = 0
        while t[0] == "name" and seen < 8:
            p = self.pkg if t[1] is None else self.pkgs.get(self.imports.get(t[1]))
            if p is None or t[2] not in p.types:
                break
            t = p.types[t[2]][2]
            seen += 1
        return t

    def complit(self, e, heap):
        ty, elems = e[1], e[2]
        return self.complit_val(ty, elems, heap)

    def complit_val(self, ty, elems, heap):
        ct = None
        u = self.underlying(ty)
        v = self.newtmp("v")
        lines = []
        if u[0] == "struct":
            ct = self.ctype(ty)
            fields = self.struct_fields(ty) if ty[0] == "name" else [fn for fn, _ in u[1]]
            ftypes = {}
            for fn, ft in u[1]:
                nm = fn if fn is not None else (ft[1] if ft[0] == "ptr" else ft)[2]
                ftypes[nm] = ft
            for i, (k, val) in enumerate(elems):
                if k is not None:
                    if k[0] != "ident":
                        raise Unsupported("struct literal key")
                    fname = k[1]
                else:
                    fname = fields[i]
                if fname in self.pkg.embedded_names and fname not in self.pkg.field_names:
                    lines.append("static_cast<%s&>(%s) = %s;" % (self.ctype(ftypes[fname]), v, self.elem_val(ftypes.get(fname), val)))
                else:
                    lines.append("%s.%s = %s;" % (v, mangle(fname), self.elem_val(ftypes.get(fname), val)))
        elif u[0] == "array":
            et = u[2]
            n = None
            if u[1] is None:
                # [...]T{...}: the length is the largest index + 1
                cnt, mx = 0, 0
                for k, val in elems:
                    if k is not None:
                        if k[0] != "int":
                            raise Unsupported("[...]T literal with a non-literal key")
                        cnt = go_int(k[1])
                    cnt += 1
                    mx = max(mx, cnt)
                ct = "Array<%s, %d>" % (self.ctype(et), mx)
            else:
                ct = self.ctype(ty)
            pos = None
            for k, val in elems:
                if k is not None:
                    pos = self.ex(k)
                    idx = pos
                    nxt = "(%s + K(1LL))" % pos
                else:
                    idx = pos if pos is not None else "K(0LL)"
                    nxt = "(%s + K(1LL))" % idx
                lines.append("%s[%s] = %s;" % (v, idx, self.elem_val(et, val)))
                pos = nxt
        elif u[0] == "slice":
            et = u[1]
            ct = self.ctype(ty)
            if any(k is not None for k, _ in elems):
                raise Unsupported("keyed slice literal")
            vals = ", ".join("%s(%s)" % (self.ctype(et), self.elem_val(et, val)) for _, val in elems)
            expr = "%s(Slice<%s>{%s})" % (ct, self.ctype(et), vals) if ty[0] == "name" else "Slice<%s>{%s}" % (self.ctype(et), vals)
            if heap:
                return "(new (go::rt::tag) %s(%s))" % (ct, expr)
            return expr
        else:
            raise Unsupported("composite literal of %r" % (u[0],))
        body = " ".join(lines)
        cap = "&" if self.scopes else ""  # (package-level initialisers: a non-local lambda takes no capture default)
        if heap:
            return "([%s]{ auto* %s_p = new (go::rt::tag) %s(); auto& %s = *%s_p; %s return %s_p; }())" % (cap, v, ct, v, v, body, v)
        return "([%s]{ %s %s{}; %s return %s; }())" % (cap, ct, v, body, v)

    def elem_val(self, et, val):
        if val[0] == "litval":
            if et is None:
                raise Unsupported("elided literal type without element type")
            if et[0] == "ptr":
                return self.complit_val(et[1], val[1], True)
            return self.complit_val(et, val[1], False)
        return self.ex(val)

    # ---- statements ----
    def block_body(self, blk):
        for s in blk[1]:
            self.stmt(s)

    def stmt(self, s):
        k = s[0]
        m = getattr(self, "st_" + k, None)
        if m is None:
            raise Unsupported("statement %r" % (k,))
        try:
            m(s)
        except Unsupported as u:
            if "(line" not in str(u) and isinstance(s[-1], int):
                raise Unsupported("%s (line %d)" % (u, s[-1]))
            raise

    def st_block(self, s):
        self.w("{")
        self.ind += 1
        self.push()
        self.block_body(s)
        self.pop()
        self.ind -= 1
        self.w("}")

    def st_exprstmt(self, s):
        self.w("%s;" % self.ex(s[1]))

    def st_incdec(self, s):
        self.w("%s%s;" % (self.ex(s[2]), s[1]))

    def st_opassign(self, s):
        op, lhs, rhs = s[1], self.ex(s[2]), self.ex(s[3])
        if op == "&^":
            self.w("%s &= ~(%s);" % (lhs, rhs))
        else:
            self.w("%s %s= %s;" % (lhs, op, rhs))

    def st_declstmt(self, s):
        for d in s[1]:
            if d[0] == "var":
                self.var_decl(d, local=True)
            elif d[0] == "const":
                self.const_decl(d, local=True)
            elif d[0] == "typedecl":
                # a function-local type: emitted in place (C++ allows local classes), known to the translator until the function ends
                if d[1] in self.pkg.types:
                    raise Unsupported("local type %s shadows a package type" % d[1])
                self.pkg.types[d[1]] = d
                self.local_types.append(d[1])
                self.emit_type(self.pkg, d)

    def var_decl(self, d, local):
        names, ty, vals = d[1], d[2], d[3]
        pre = "" if local else "static "
        if vals is None:
            ct = self.ctype(ty)
            for n in names:
                if local and (n in self.heap_vars or (ty is not None and ty[0] == "array" and n in self.sliced_vars)):
                    # its address is taken, or it is an array that is sliced (the slice may outlive the block: frameHeader.appendTo's
                    # `tmp[:1]` is used after tmp's scope): on the heap, like Go's escape analysis puts it
                    self.w("auto& %s = *new (go::rt::tag) %s();" % (mangle(n), ct))
                else:
                    self.w("%s%s %s{};" % (pre, ct, mangle(n)))
                if local:
                    self.declare(n)
            return
        if len(vals) == len(names):
            for n, v in zip(names, vals):
                val = self.ex(v)
                if n == "_":
                    self.w("(void)(%s);" % val)
                    continue
                if local and n in self.heap_vars:
                    if ty is not None:
                        self.w("auto& %s = *new (go::rt::tag) %s(%s);" % (mangle(n), self.ctype(ty), val))
                    else:
                        self.w("auto& %s = *new (go::rt::tag) auto(go::def(%s));" % (mangle(n), val))
                elif ty is not None:
                    self.w("%s%s %s = %s(%s);" % (pre, self.ctype(ty), mangle(n), self.ctype(ty), val) if ty[0] == "name" else "%s%s %s = %s;" % (pre, self.ctype(ty), mangle(n), val))
                else:
                    self.w("%sauto %s = go::def(%s);" % (pre, mangle(n), val))
            if local:
                for n in names:
                    self.declare(n)
            return
        if len(vals) == 1:
            t = self.newtmp()
            self.w("%sauto %s = %s;" % (pre, t, self.ex(vals[0])))
            for i, n in enumerate(names):
                if n != "_":
                    if ty is not None:
                        self.w("%s%s %s = std::get<%d>(%s);" % (pre, self.ctype(ty), mangle(n), i, t))
                    else:
                        self.w("%sauto %s = std::get<%d>(%s);" % (pre, mangle(n), i, t))
                    if local:
                        self.declare(n)
            return
        raise Unsupported("var declaration shape")

    def const_decl(self, d, local):
        names, ty, vals, idx = d[1], d[2], d[3], d[4]
        self.iota = idx
        for n, v in zip(names, vals):
            if n == "_":
                continue
            val = self.ex(v)
            if ty is not None:
                ct = self.ctype(ty)
                if ct == "String":
                    self.w("static const String %s = %s;" % (mangle(n), val))
                elif ct in ("float64", "float32"):
                    self.w("static constexpr %s %s = %s;" % (ct, mangle(n), val))
                else:
                    self.w("static const %s %s = %s(%s);" % (ct, mangle(n), ct, val))
            else:
                if v[0] == "str" or "String(" in val:
                    self.w("static const String %s = %s;" % (mangle(n), val))
                elif v[0] in ("ident",) and v[1] in ("true", "false"):
                    self.w("static constexpr bool %s = %s;" % (mangle(n), val))
                elif "go::len(" in val:
                    self.w("static const auto %s = %s;" % (mangle(n), val))
                else:
                    self.w("static constexpr auto %s = %s;" % (mangle(n), val))
            if local:
                self.declare(n)

    def st_define(self, s):
        lhs, rhs = s[1], s[2]
        names = []
        for l in lhs:
            if l[0] != "ident":
                raise Unsupported(":= with a non-identifier on the left")
            names.append(l[1])
        cur = self.scopes[-1]
        if len(lhs) == len(rhs):
            if len(lhs) == 1:
                n = names[0]
                val = self.rhs_value(rhs[0])
                if n == "_":
                    self.w("(void)(%s);" % val)
                elif n in cur:
                    self.w("%s = %s;" % (mangle(n), val))
                elif n in self.heap_vars:
                    used = set()
                    self.idents_in(rhs[0], used)
                    if n in used:  # `br := &br[i]`: the right side still means the outer br
                        t = self.newtmp()
                        self.w("auto %s = go::def(%s);" % (t, val))
                        self.w("auto& %s = *new (go::rt::tag) auto(%s);" % (mangle(n), t))
                    else:
                        self.w("auto& %s = *new (go::rt::tag) auto(go::def(%s));" % (mangle(n), val))
                    self.declare(n)
                else:
                    used = set()
                    self.idents_in(rhs[0], used)
                    if n in used:  # `s := s + 1`: the right side still means the outer s
                        t = self.newtmp()
                        self.w("auto %s = go::def(%s);" % (t, val))
                        self.w("auto %s = %s;" % (mangle(n), t))
                    else:
                        self.w("auto %s = go::def(%s);" % (mangle(n), val))
                    self.declare(n)
                return
            temps = []
            for r in rhs:
                t = self.newtmp()
                self.w("auto %s = go::def(%s);" % (t, self.rhs_value(r)))
                temps.append(t)
            for n, t in zip(names, temps):
                if n == "_":
                    continue
                if n in cur:
                    self.w("%s = %s;" % (mangle(n), t))
                elif n in self.heap_vars:
                    self.w("auto& %s = *new (go::rt::tag) auto(%s);" % (mangle(n), t))
                    self.declare(n)
                else:
                    self.w("auto %s = %s;" % (mangle(n), t))
                    self.declare(n)
            return
        if len(rhs) == 1:
            t = self.newtmp()
            self.w("auto %s = %s;" % (t, self.ex(rhs[0])))
            for i, n in enumerate(names):
                if n == "_":
                    continue
                if n in cur:
                    self.w("%s = std::get<%d>(%s);" % (mangle(n), i, t))
                else:
                    self.w("auto %s = std::get<%d>(%s);" % (mangle(n), i, t))
                    self.declare(n)
            return
        raise Unsupported("define shape")

    def rhs_value(self, r):
        if r[0] == "funclit":
            return self.ex_funclit(r, "&")
        return self.ex(r)

    def st_assign(self, s):
        lhs, rhs = s[1], s[2]
        if len(lhs) == len(rhs):
            if len(lhs) == 1:
                if lhs[0] == ("ident", "_"):
                    self.w("(void)(%s);" % self.ex(rhs[0]))
                else:
                    self.w("%s = %s;" % (self.ex(lhs[0]), self.rhs_value(rhs[0])))
                return
            self.w("{")
            self.ind += 1
            temps = []
            for r in rhs:
                t = self.newtmp()
                self.w("auto %s = go::def(%s);" % (t, self.ex(r)))
                temps.append(t)
            for l, t in zip(lhs, temps):
                if l != ("ident", "_"):
                    self.w("%s = %s;" % (self.ex(l), t))
            self.ind -= 1
            self.w("}")
            return
        if len(rhs) == 1:
            self.w("{")
            self.ind += 1
            t = self.newtmp()
            self.w("auto %s = %s;" % (t, self.ex(rhs[0])))
            for i, l in enumerate(lhs):
                if l != ("ident", "_"):
                    self.w("%s = std::get<%d>(%s);" % (self.ex(l), i, t))
            self.ind -= 1
            self.w("}")
            return
        raise Unsupported("assign shape")

    def st_return(self, s):
        vals = s[1]
        nr = self.named_results
        if not vals:
            if nr:
                if len(nr) == 1:
                    self.w("return %s;" % mangle(nr[0]))
                else:
                    self.w("return {%s};" % ", ".join(mangle(n) for n in nr))
            else:
                self.w("return;")
            return
        if nr and len(vals) == len(nr):  # assign the named results first (deferred closures read them)
            if len(nr) == 1:
                self.w("return (%s = %s);" % (mangle(nr[0]), self.rhs_value(vals[0])))
            else:
                self.w("{")
                self.ind += 1
                temps = []
                for v in vals:
                    t = self.newtmp()
                    self.w("auto %s = go::def(%s);" % (t, self.ex(v)))
                    temps.append(t)
                for n, t in zip(nr, temps):
                    if n != "_":
                        self.w("%s = %s;" % (mangle(n), t))
                self.w("return {%s};" % ", ".join(mangle(n) if n != "_" else t for n, t in zip(nr, temps)))
                self.ind -= 1
                self.w("}")
            return
        if len(vals) == 1:
            v = vals[0]
            if v[0] == "funclit":
                self.w("return %s;" % self.ex_funclit(v, "&"))
            else:
                self.w("return %s;" % self.ex(v))
            return
        self.w("return {%s};" % ", ".join(self.ex(v) for v in vals))

    def const_bool(self, e):
        """True / False when the expression is a compile-time boolean constant of the package (debug switches), else None."""
        k = e[0]
        if k == "ident" and not self.is_local(e[1]):
            if e[1] == "true":
                return True
            if e[1] == "false":
                return False
            return self.pkg.bool_consts.get(e[1])
        if k == "paren":
            return self.const_bool(e[1])
        if k == "unary" and e[1] == "!":
            v = self.const_bool(e[2])
            return None if v is None else (not v)
        if k == "binary" and e[1] in ("&&", "||"):
            a, b = self.const_bool(e[2]), self.const_bool(e[3])
            if e[1] == "&&":
                if a is False or b is False:
                    return False
                if a is True and b is True:
                    return True
            else:
                if a is True or b is True:
                    return True
                if a is False and b is False:
                    return False
        return None

    def st_if(self, s):
        init, cond, body, els = s[1], s[2], s[3], s[4]
        if init is None and self.const_bool(cond) is False:  # `if debug { ... }`: dead code, as for the Go compiler
            if els is not None:
                if els[0] == "if":
                    self.st_if(els)
                else:
                    self.st_block(els)
            return
        self.push()
        if init is not None:
            self.w("{")
            self.ind += 1
            self.stmt(init)
        self.w("if (%s) {" % self.ex(cond))
        self.ind += 1
        self.push()
        self.block_body(body)
        self.pop()
        self.ind -= 1
        if els is None:
            self.w("}")
        else:
            self.w("} else {")
            self.ind += 1
            self.push()
            if els[0] == "if":
                self.st_if(els)
            else:
                self.block_body(els)
            self.pop()
            self.ind -= 1
            self.w("}")
        if init is not None:
            self.ind -= 1
            self.w("}")
        self.pop()

    def loop_open(self, label):
        cl = self.newtmp("c")
        bl = self.newtmp("b")
        if label:
            self.labels[label] = (cl, bl)
        self.ctx.append(("loop", cl, bl))
        return cl, bl

    def st_for(self, s):
        label, init, cond, post, body = s[1], s[2], s[3], s[4], s[5]
        if label and label in self.goto_targets:
            self.w("%s:;" % mangle(label))
        self.w("{")
        self.ind += 1
        self.push()
        if init is not None:
            self.stmt(init)
        cl, bl = self.loop_open(label)
        self.w("for (;;) {")
        self.ind += 1
        if cond is not None:
            self.w("if (!(%s)) break;" % self.ex(cond))
        self.w("{")
        self.ind += 1
        self.push()
        self.block_body(body)
        self.pop()
        self.ind -= 1
        self.w("}")
        self.w("%s:;" % cl)
        if post is not None:
            self.stmt(post)
        self.ind -= 1
        self.w("}")
        self.w("%s:;" % bl)
        self.ctx.pop()
        self.pop()
        self.ind -= 1
        self.w("}")

    def st_forrange(self, s):
        label, lhs, define, x, body = s[1], s[2], s[3], s[4], s[5]
        self.w("{")
        self.ind += 1
        self.push()
        r = self.newtmp("r")
        n = self.newtmp("n")
        i = self.newtmp("i")
        self.w("auto&& %s = %s;" % (r, self.ex(x)))
        self.w("const long long %s = go::rangelen(%s);" % (n, r))
        cl, bl = self.loop_open(label)
        self.w("for (long long %s = 0; %s < %s; %s++) {" % (i, i, n, i))
        self.ind += 1
        self.push()
        key = lhs[0] if len(lhs) > 0 else None
        val = lhs[1] if len(lhs) > 1 else None
        if key is not None and key != ("ident", "_"):
            if define:
                self.w("Int %s = Int::raw(%s);" % (mangle(key[1]), i))
                self.declare(key[1])
            else:
                self.w("%s = Int::raw(%s);" % (self.ex(key), i))
        if val is not None and val != ("ident", "_"):
            if define:
                self.w("auto %s = go::rangeval(%s, %s);" % (mangle(val[1]), r, i))
                self.declare(val[1])
            else:
                self.w("%s = go::rangeval(%s, %s);" % (self.ex(val), r, i))
        self.w("{")
        self.ind += 1
        self.push()
        self.block_body(body)
        self.pop()
        self.ind -= 1
        self.w("}")
        self.w("%s:;" % cl)
        self.pop()
        self.ind -= 1
        self.w("}")
        self.w("%s:;" % bl)
        self.ctx.pop()
        self.pop()
        self.ind -= 1
        self.w("}")

    def st_switch(self, s):
        label, init, tag, cases = s[1], s[2], s[3], s[4]
        self.w("{")
        self.ind += 1
        self.push()
        if init is not None:
            self.stmt(init)
        end = self.newtmp("s")
        if label:
            self.labels[label] = (None, end)
        self.ctx.append(("switch", None, end))
        tv = None
        if tag is not None:
            tv = self.newtmp("g")
            self.w("auto %s = go::def(%s);" % (tv, self.ex(tag)))
        first = True
        default = None
        for exprs, body in cases:
            if any(st[0] == "fallthrough" for st in body):
                raise Unsupported("fallthrough")
            if exprs is None:
                default = body
                continue
            if tv is not None:
                cond = " || ".join("(%s == %s)" % (tv, self.ex(x)) for x in exprs)
            else:
                cond = " || ".join("(%s)" % self.ex(x) for x in exprs)
            self.w("%sif (%s) {" % ("" if first else "} else ", cond))
            first = False
            self.ind += 1
            self.push()
            for st in body:
                self.stmt(st)
            self.pop()
            self.ind -= 1
        if default is not None:
            self.w("{" if first else "} else {")
            self.ind += 1
            self.push()
            for st in default:
                self.stmt(st)
            self.pop()
            self.ind -= 1
            self.w("}")
        elif not first:
            self.w("}")
        self.w("%s:;" % end)
        self.ctx.pop()
        self.pop()
        self.ind -= 1
        self.w("}")

    def st_break(self, s):
        label = s[1]
        if label:
            self.w("goto %s;" % self.labels[label][1])
            return
        top = self.ctx[-1]
        if top[0] == "switch":
            self.w("goto %s;" % top[2])
        else:
            self.w("goto %s;" % top[2])

    def st_continue(self, s):
        label = s[1]
        if label:
            self.w("goto %s;" % self.labels[label][0])
            return
        for c in reversed(self.ctx):
            if c[0] == "loop":
                self.w("goto %s;" % c[1])
                return
        raise Unsupported("continue outside a loop")

    def st_goto(self, s):
        self.w("goto %s;" % mangle(s[1]))

    def st_labeled(self, s):
        self.w("%s:;" % mangle(s[1]))
        if s[2] is not None:
            self.stmt(s[2])

    def st_defer(self, s):
        call = s[1]
        d = self.newtmp("d")
        if call[0] == "call" and call[1][0] == "funclit" and not call[2]:
            self.w("go::Defer %s(%s);" % (d, self.ex_funclit(call[1], "&")))
        else:
            self.w("go::Defer %s([&] { %s; });" % (d, self.ex(call)))

    def st_go(self, s):
        # `go f(x)`: run to completion where it is started.  A valid schedule of the program wherever the goroutine never waits for
        # something its starter does later — the encoder's two (encoder.go nextBlock: block encode, block write) only wait for the
        # PREVIOUS block's goroutines through sync.WaitGroup, which have then already finished — and the only one a translation
        # without a scheduler can offer; files whose goroutines talk through channels (enc_jobs.go) are not translated.
        self.w("%s;  // go" % self.ex(s[1]))

    def st_send(self, s):
        self.w("go::send(%s, %s);" % (self.ex(s[1]), self.ex(s[2])))

    # ---- functions ----
    def func_prologue(self, sig):
        results = sig[2]
        self.named_results = None
        if results and results[0][0] is not None:
            self.named_results = [r[0] for r in results]
            for r in results:
                if r[0] != "_":
                    self.w("%s %s{};" % (self.ctype(r[1]), mangle(r[0])))
                    self.declare(r[0])
        self.cur_results = results

    def params_text(self, sig, boxed=False, heap=()):
        out = []
        for p in sig[1]:
            n = mangle(p[0]) + ("__arg" if (boxed or p[0] in heap) else "") if p[0] and p[0] != "_" else self.newtmp("u")
            out.append("%s %s" % (self.param_type(p), n))
        return ", ".join(out)

    def emit_func_body(self, f, qual):
        name, recv, sig, body, tparams = f[1], f[2], f[3], f[4], f[5]
        self.push()
        self.ctx, self.labels = [], {}
        tmpl = ""
        if tparams:
            tmpl = "template <%s> " % ", ".join("class %s" % mangle(n) for n, _ in tparams)
            for n, _ in tparams:
                self.declare(n)
        for p in sig[1]:
            if p[0]:
                self.declare(p[0])
        boxed = any(st[0] == "return" and len(st[1]) == 1 and st[1][0][0] == "funclit" for st in body[1])
        # inside a method, C++ finds the class's members before the namespace's functions: a package-level function that shares its
        # name with a method of the receiver's type (s2: encodeBlock) has to be written with its namespace
        self.recv_methods = set()
        if recv is not None:
            rt = recv[1][1] if recv[1][0] == "ptr" else recv[1]
            self.recv_methods = {m[1][1] for m in self.pkg.methods.get(rt[2], [])}
        self.heap_vars = set()
        self.addr_roots(body, self.heap_vars)
        self.sliced_vars = set()
        self.slice_roots(body, self.sliced_vars)
        self.goto_targets = set()
        self.gotos_in(body, self.goto_targets)
        self.w("%s%s %s%s(%s) {" % (tmpl, self.result_type(sig[2]), qual, mangle(name), self.params_text(sig, boxed, self.heap_vars)))
        self.ind += 1
        if recv is not None and recv[0] and recv[0] != "_":
            if recv[1][0] == "ptr":
                self.w("auto* %s = this; (void)%s;" % (mangle(recv[0]), mangle(recv[0])))
            else:
                self.w("auto %s = *this; (void)%s;" % (mangle(recv[0]), mangle(recv[0])))
            self.declare(recv[0])
        for p in sig[1]:  # parameters a returned closure captures, or whose address is taken: they may outlive this call (Go: escape to the heap)
            if p[0] and p[0] != "_" and (boxed or p[0] in self.heap_vars):
                self.w("auto& %s = *new (go::rt::tag) %s(%s__arg);" % (mangle(p[0]), self.param_type(p), mangle(p[0])))
        self.func_prologue(sig)
        self.local_types = []
        self.block_body(body)
        for n in self.local_types:
            del self.pkg.types[n]
        self.local_types = []
        self.ind -= 1
        self.w("}")
        self.pop()

    # ---- package ----
    def collect(self, pkg):
        for path, ast in pkg.files:
            skip = self.cfg.get("skip", {}).get(path, set())
            only = self.cfg.get("only", {}).get(path)
            for d in ast[3]:
                k = d[0]
                if only is not None:  # this file contributes the listed declarations only
                    if k == "typedecl" and d[1] not in only:
                        continue
                    if k == "func":
                        if d[2] is None and d[1] not in only:
                            continue
                        if d[2] is not None:
                            rt0 = d[2][1][1] if d[2][1][0] == "ptr" else d[2][1]
                            if "%s.%s" % (rt0[2], d[1]) not in only and rt0[2] + ".*" not in only:
                                continue
                    if k in ("const", "var") and not any(n in only for n in d[1]):
                        continue
                if k == "typedecl":
                    if d[1] in skip:
                        continue
                    pkg.types[d[1]] = d
                elif k == "func":
                    name, recv = d[1], d[2]
                    if recv is None:
                        if name in skip:
                            continue
                        if name == "init":
                            pkg.inits.append((path, d))
                        else:
                            pkg.funcs[name] = d
                    else:
                        rt = recv[1][1] if recv[1][0] == "ptr" else recv[1]
                        rname = rt[2]
                        if "%s.%s" % (rname, name) in skip or rname in skip:
                            continue
                        pkg.methods.setdefault(rname, []).append((path, d))
                        pkg.method_names.add(name)
                elif k == "const":
                    if all(n in skip for n in d[1]):
                        continue
                    pkg.consts.append((path, d))
                elif k == "var":
                    if all(n in skip for n in d[1]):
                        continue
                    pkg.vars.append((path, d))
                    pkg.var_names |= set(d[1])
        for td in pkg.types.values():
            if td[2][0] == "struct":
                for fn, ft in td[2][1]:
                    if fn is not None:
                        pkg.field_names.add(fn)
                    else:
                        pkg.embedded_names.add((ft[1] if ft[0] == "ptr" else ft)[2])
        for path, ast in pkg.files:
            for d in ast[3]:
                if d[0] == "func":
                    self.bound_names(d[3], pkg.local_names)
                    self.bound_names(d[4], pkg.local_names)
                    if d[2] is not None and d[2][0]:
                        pkg.local_names.add(d[2][0])
        self.pkg = pkg
        for _ in range(3):  # (debugEncoder = debug, debugAsserts = debug || false ...)
            for path, d in pkg.consts:
                if d[2] is None and d[3] is not None and len(d[1]) == len(d[3]):
                    for n, v in zip(d[1], d[3]):
                        b = self.const_bool(v)
                        if b is not None:
                            pkg.bool_consts[n] = b

    def type_deps(self, t, by_value=True):
        """Names of package types a type needs COMPLETE (held by value)."""
        k = t[0]
        if k == "name":
            return {t[2]} if t[1] is None and by_value else set()
        if k == "array":
            return self.type_deps(t[2], by_value)
        if k == "struct":
            out = set()
            for fn, ft in t[1]:
                out |= self.type_deps(ft, True)
            return out
        return set()

    def emit_package(self, pkg):
        self.pkg = pkg
        self.all_methods = {}
        self.all_method_names = set()
        self.all_field_names = set()
        for p in self.pkgs.values():
            self.all_method_names |= p.method_names
            self.all_field_names |= p.field_names
            for rname, ms in p.methods.items():
                for path, f in ms:
                    self.all_methods.setdefault(f[1], []).append(f)
        ns = mangle(pkg.name)
        self.w("namespace %s {" % ns)
        self.w("using namespace go;")
        file_imports = {path: {n: p.split("/")[-1] for n, p in ast[2]} for path, ast in pkg.files}
        self.scopes = []
        # forward declarations
        for n, td in pkg.types.items():
            if td[2][0] != "functype":
                self.w("struct %s;" % self.tname(pkg, n))
        # constants first (array lengths use them), in source order; typed constants of package types come after the types
        late_consts = []
        for path, d in self.order_consts(pkg):
            self.imports = file_imports[path]
            if d[2] is not None and d[2][0] == "name" and d[2][1] is None and d[2][2] in pkg.types:
                late_consts.append((path, d))
                continue
            try:
                self.const_decl(d, local=False)
            except Unsupported as u:
                self.warnings.append("%s: const %s: %s" % (path, d[1], u))
        # types in dependency order
        done, order = set(), []

        def visit(n, stack=()):
            if n in done or n not in pkg.types:
                return
            if n in stack:
                return
            for dep in sorted(self.type_deps(pkg.types[n][2])):
                visit(dep, stack + (n,))
            done.add(n)
            order.append(n)
        for n in pkg.types:
            visit(n)
        for n in order:
            td = pkg.types[n]
            path = next(p for p, ast in pkg.files if td in ast[3])
            self.imports = file_imports[path]
            self.emit_type(pkg, td)
        for path, d in late_consts:
            self.imports = file_imports[path]
            self.const_decl(d, local=False)
        # function prototypes
        for n, f in pkg.funcs.items():
            path = next(p for p, ast in pkg.files if f in ast[3])
            self.imports = file_imports[path]
            if f[5]:
                continue  # templates are defined before use, below
            self.push()
            self.w("%s %s(%s);" % (self.result_type(f[3][2]), mangle(n), self.params_text(f[3])))
            self.pop()
        # package variables
        for path, d in pkg.vars:
            self.imports = file_imports[path]
            try:
                self.var_decl(d, local=False)
            except Unsupported as u:
                self.warnings.append("%s: var %s: %s" % (path, d[1], u))
        # bodies
        for n, f in pkg.funcs.items():
            if f[4] is None:
                self.emit_asm_thunk(n, f)
                continue
            path = next(p for p, ast in pkg.files if f in ast[3])
            self.imports = file_imports[path]
            self.emit_func_body(f, "")
        for rname, ms in pkg.methods.items():
            if rname not in pkg.types:
                continue
            for path, f in ms:
                if f[4] is None:
                    continue
                self.imports = file_imports[path]
                self.emit_func_body(f, self.tname(pkg, rname) + "::")
        # init functions
        self.w("inline void go_init() {")
        self.ind += 1
        self.w("static bool done = false; if (done) return; done = true;")
        self.ind -= 1
        for i, (path, f) in enumerate(pkg.inits):
            self.imports = file_imports[path]
            self.ind += 1
            self.w("{")
            self.ind += 1
            self.push()
            self.ctx, self.labels, self.named_results = [], {}, None
            self.block_body(f[4])
            self.pop()
            self.ind -= 1
            self.w("}")
            self.ind -= 1
        self.w("}")
        self.w("}  // namespace %s" % ns)
        self.w()

    def emit_asm_thunk(self, n, f):
        """A Go function declared without a body is implemented in the package's assembly (TEXT ·name).  oracle/ref_s2asm/
        plan9_to_gas.py re-spells that assembly as p9_name(uint64_t* frame), `frame` being the function's argument frame in Go's
        stack-based ABI0: arguments, then results, in declaration order, each at its natural alignment (a slice is base / len / cap)."""
        sig = f[3]
        words, off = [], 0

        def kind(ty):
            if ty[0] == "ptr":
                return "ptr"
            if ty[0] == "slice":
                return "slice"
            if ty[0] == "name" and ty[1] is None and ty[2] in ("int", "uint", "int64", "uint64", "uintptr"):
                return "word"
            if ty[0] == "name" and ty[1] is None and ty[2] == "bool":
                return "bool"
            raise Unsupported("assembly function %s: parameter type %r" % (n, ty))
        self.push()
        names = []
        for p in sig[1]:
            pn = mangle(p[0])
            names.append(pn)
            k = kind(p[1])
            if k == "ptr":
                words.append("(uint64_t)(uintptr_t)%s" % pn)
            elif k == "word":
                words.append("(uint64_t)%s.v" % pn)
            elif k == "slice":
                words += ["(uint64_t)(uintptr_t)%s.p" % pn, "(uint64_t)%s.n" % pn, "(uint64_t)%s.c" % pn]
            else:
                raise Unsupported("assembly function %s: bool argument" % n)
        nargs = len(words)
        res = sig[2]
        if len(res) > 1:
            raise Unsupported("assembly function %s: several results" % n)
        self.w('extern "C" void p9_%s(uint64_t* frame);' % n)
        self.w("%s %s(%s) {" % (self.result_type(res), mangle(n), self.params_text(sig)))
        self.ind += 1
        self.w("uint64_t frame__[%d] = {%s};" % (nargs + 1, ", ".join(words)))
        self.w("p9_%s(frame__);" % n)
        if res:
            k = kind(res[0][1])
            if k == "word":
                self.w("return %s::raw((%s::raw_type)frame__[%d]);" % (self.ctype(res[0][1]), self.ctype(res[0][1]), nargs))
            elif k == "bool":
                self.w("return (frame__[%d] & 0xFF) != 0;" % nargs)
            else:
                raise Unsupported("assembly function %s: result type" % n)
        self.ind -= 1
        self.w("}")
        self.pop()

    def bound_names(self, e, out):
        """Every name a function binds locally (parameters, :=, var, range): a type of the same name needs another C++ name."""
        if isinstance(e, tuple):
            if e and e[0] in ("define", "forrange") and isinstance(e[1 if e[0] == "define" else 2], list):
                for l in e[1 if e[0] == "define" else 2]:
                    if l[0] == "ident":
                        out.add(l[1])
            if e and e[0] == "var":
                out |= set(e[1])
            if e and e[0] == "sig":
                for p in e[1] + e[2]:
                    if p[0]:
                        out.add(p[0])
            for x in e:
                self.bound_names(x, out)
        elif isinstance(e, list):
            for x in e:
                self.bound_names(x, out)

    def gotos_in(self, e, out):
        if isinstance(e, tuple):
            if len(e) == 3 and e[0] == "goto":
                out.add(e[1])
            for x in e:
                self.gotos_in(x, out)
        elif isinstance(e, list):
            for x in e:
                self.gotos_in(x, out)

    def addr_roots(self, e, out):
        """Names of variables whose address is taken somewhere in e (&x, &x.f, &x[i]): Go moves such locals to the heap."""
        if isinstance(e, tuple):
            if len(e) == 3 and e[0] == "unary" and e[1] == "&":
                r = e[2]
                while r[0] in ("selector", "index", "paren"):
                    r = r[1]
                if r[0] == "ident":
                    out.add(r[1])
            for x in e:
                self.addr_roots(x, out)
        elif isinstance(e, list):
            for x in e:
                self.addr_roots(x, out)

    def slice_roots(self, e, out):
        """Names of variables that are sliced somewhere in e (x[a:b]); the local ARRAYS among them go to the heap (var_decl)."""
        if isinstance(e, tuple):
            if len(e) == 6 and e[0] == "slice":
                r = e[1]
                while r[0] in ("selector", "index", "paren"):
                    r = r[1]
                if r[0] == "ident":
                    out.add(r[1])
            for x in e:
                self.slice_roots(x, out)
        elif isinstance(e, list):
            for x in e:
                self.slice_roots(x, out)

    def idents_in(self, e, out):
        if isinstance(e, tuple):
            if e and e[0] == "ident":
                out.add(e[1])
            for x in e:
                self.idents_in(x, out)
        elif isinstance(e, list):
            for x in e:
                self.idents_in(x, out)

    def order_consts(self, pkg):
        """Constant declarations in an order where every constant follows the ones its value mentions (Go has no such rule)."""
        by_name = {}
        for item in pkg.consts:
            for n in item[1][1]:
                by_name[n] = item
        done, out = set(), []

        def visit(item, depth=0):
            if id(item) in done or depth > 50:
                return
            done.add(id(item))
            used = set()
            self.idents_in(item[1][3], used)
            self.idents_in(item[1][2], used)
            for u in sorted(used):
                if u in by_name and by_name[u] is not item:
                    visit(by_name[u], depth + 1)
            out.append(item)
        for item in pkg.consts:
            visit(item)
        return out

    def method_decls(self, pkg, name):
        out = []
        for path, f in pkg.methods.get(name, []):
            self.push()
            out.append("%s %s(%s);" % (self.result_type(f[3][2]), mangle(f[1]), self.params_text(f[3])))
            self.pop()
        return out

    def emit_type(self, pkg, td):
        name, ty = td[1], td[2]
        n = self.tname(pkg, name)
        if ty[0] == "struct":
            bases = []
            fields = []
            for fn, ft in ty[1]:
                if fn is None:
                    if ft[0] == "ptr":
                        raise Unsupported("embedded pointer in %s" % name)
                    bases.append(self.ctype(ft))
                else:
                    fields.append((fn, ft))
            drop = self.cfg.get("drop_fields", {}).get("%s.%s" % (pkg.name, name), set())
            self.w("struct %s%s {" % (n, (" : " + ", ".join(bases)) if bases else ""))
            self.ind += 1
            for fn, ft in fields:
                if fn in drop:
                    continue
                if fn == "_":
                    continue
                self.w("%s %s{};" % (self.ctype(ft), mangle(fn)))
            self.w("%s* operator->() { return this; }" % n)
            self.w("const %s* operator->() const { return this; }" % n)
            for m in self.method_decls(pkg, name):
                self.w(m)
            self.ind -= 1
            self.w("};")
            return
        if ty[0] == "interface":
            self.emit_interface(pkg, name, ty)
            return
        if ty[0] == "functype":
            self.w("typedef %s %s;" % (self.ctype(ty), n))
            return
        # a named non-struct type: derive from the underlying representation so that methods can be members
        base = self.ctype(ty)
        self.w("struct %s : %s {" % (n, base))
        self.ind += 1
        self.w("typedef %s Base_;" % base)
        self.w("using Base_::Base_;")
        self.w("%s() {}" % n)
        self.w("%s(const Base_& b) : Base_(b) {}" % n)
        self.w("%s* operator->() { return this; }" % n)
        self.w("const %s* operator->() const { return this; }" % n)
        for m in self.method_decls(pkg, name):
            self.w(m)
        self.ind -= 1
        self.w("};")

    def emit_interface(self, pkg, name, ty):
        n = self.tname(pkg, name)
        methods = ty[1]
        self.w("struct %s {" % n)
        self.ind += 1
        self.w("struct Base_ {")
        self.ind += 1
        self.w("virtual ~Base_() {}")
        self.w("virtual void* obj_() = 0;")
        self.w("virtual const void* tid_() = 0;")  # the dynamic type, for type assertions (go::cast)
        sigs = []
        for mn, sig in methods:
            self.push()
            ps = self.params_text(sig)
            self.pop()
            names = [p.split(" ")[-1] for p in ps.split(", ")] if ps else []
            sigs.append((mangle(mn), self.result_type(sig[2]), ps, ", ".join(names)))
        for mn, rt, ps, an in sigs:
            self.w("virtual %s %s(%s) = 0;" % (rt, mn, ps))
        self.ind -= 1
        self.w("};")
        self.w("template <class T_> struct Impl_ : Base_ {")
        self.ind += 1
        self.w("T_* p;")
        self.w("explicit Impl_(T_* q) : p(q) {}")
        self.w("void* obj_() override { return (void*)p; }")
        self.w("const void* tid_() override { return go::type_tag<T_>(); }")
        for mn, rt, ps, an in sigs:
            self.w("%s %s(%s) override { return p->%s(%s); }" % (rt, mn, ps, mn, an))
        self.ind -= 1
        self.w("};")
        self.w("Base_* b_ = nullptr;")
        self.w("%s() {}" % n)
        self.w("%s(go::Nil) {}" % n)
        self.w("template <class T_> %s(T_* p) : b_(p ? new (go::rt::tag) Impl_<T_>(p) : nullptr) {}" % n)
        self.w("%s* operator->() { return this; }" % n)
        for mn, rt, ps, an in sigs:
            self.w("%s %s(%s) { return b_->%s(%s); }" % (rt, mn, ps, mn, an))
        self.w("friend bool operator==(const %s& a, go::Nil) { return a.b_ == nullptr; }" % n)
        self.w("friend bool operator!=(const %s& a, go::Nil) { return a.b_ != nullptr; }" % n)
        self.ind -= 1
        self.w("};")


def load_package(root, name, rel_files, patches):
    files = []
    for rel in rel_files:
        src = open(os.path.join(root, rel), encoding="utf-8").read()
        for pat, rep, why in patches.get(rel, []):
            new = re.sub(pat, rep, src, flags=re.S)
            if new == src:
                raise SystemExit("go2cpp: patch for %s did not apply (%s)" % (rel, why))
            src = new
        files.append((rel, goparse.parse(src, rel)))
    return Package(name, files)


def translate(root, cfg):
    pkgs = [load_package(root, name, files, cfg.get("patches", {})) for name, files in cfg["packages"]]
    em = Emitter(pkgs, cfg)
    for p in pkgs:
        em.pkg = p
        em.collect(p)
    em.w("// GENERATED at build time by oracle/ref_go/go2cpp.py from the reference's Go sources — not part of the repository.")
    em.w('#include "gort.h"')
    em.w()
    for p in pkgs:
        em.emit_package(p)
    return "\n".join(em.out) + "\n", em.warnings


if __name__ == "__main__":
    import manifest
    root, out = sys.argv[1], sys.argv[2]
    flavour = sys.argv[3] if len(sys.argv) > 3 else "generic"
    text, warnings = translate(root, manifest.CFG if flavour == "generic" else manifest.flavour(flavour))
    with open(out, "w") as f:
        f.write(text)
    for wmsg in warnings:
        sys.stderr.write("go2cpp: " + wmsg + "\n")
    sys.stderr.write("go2cpp: wrote %s (%d lines)\n" % (out, text.count("\n")))
