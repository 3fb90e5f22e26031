#!/usr/bin/env python3
"""Build recipe of oracle/_ref (test infrastructure): the reference's OWN amd64 S2 block encoders, runnable without a Go toolchain.

klauspost/compress ships its S2 block encoders for amd64 as generated Go (Plan 9) assembly, s2/encodeblock_amd64.s
(s2/_generate/gen.go) — this is what every amd64 user of s2.Encode / EncodeBetter / EncodeSnappy / EncodeSnappyBetter runs
(s2/encode_amd64.go).  The file uses ~50 mnemonics in a regular way, so it can be re-spelt for the GNU assembler line by line:

    python3 plan9_to_gas.py /root/reference/s2/encodeblock_amd64.s oracle/_ref/s2ref_amd64.S

Nothing of the reference is copied into the repository: the output goes to oracle/_ref/ (git-ignored) and is assembled there
(oracle/Makefile, target _ref).  Each TEXT symbol ·name becomes p9_name(uint64_t* frame): `frame` is the function's Go argument
frame (ABI0: arguments and results in memory, in declaration order, slices as base/len/cap), copied onto the stack where the
body expects it (name+off(FP)), and copied back on return so that the result slots can be read.  The bodies are unchanged
instruction for instruction; only the spelling differs:

  * operand order is kept (Plan 9 and AT&T both write source first) except for CMPx, whose operands Plan 9 writes the Intel way;
  * register width comes from the mnemonic suffix (MOVL AX -> %eax), address registers are always 64-bit;
  * off(SP) are the function's locals, name+off(FP) its arguments: both become offsets from %rsp in a frame laid out here;
  * the `#ifdef GOAMD64_v3` alternatives take their #else branch (BSFQ instead of TZCNTQ: same result for the non-zero inputs
    they are given), like a default GOAMD64=v1 build.
"""
import re
import sys

REG64 = {"AX": "rax", "BX": "rbx", "CX": "rcx", "DX": "rdx", "SI": "rsi", "DI": "rdi", "BP": "rbp", "SP": "rsp"}
REG32 = {"AX": "eax", "BX": "ebx", "CX": "ecx", "DX": "edx", "SI": "esi", "DI": "edi", "BP": "ebp"}
REG16 = {"AX": "ax", "BX": "bx", "CX": "cx", "DX": "dx", "SI": "si", "DI": "di", "BP": "bp"}
REG8 = {"AX": "al", "BX": "bl", "CX": "cl", "DX": "dl", "SI": "sil", "DI": "dil", "BP": "bpl"}
for i in range(8, 16):
    REG64["R%d" % i] = "r%d" % i
    REG32["R%d" % i] = "r%dd" % i
    REG16["R%d" % i] = "r%dw" % i
    REG8["R%d" % i] = "r%db" % i
BYSIZE = {8: REG64, 4: REG32, 2: REG16, 1: REG8}

# mnemonic -> (gas mnemonic, operand size in bytes or None for SSE / jumps)
OPS = {}
for base, gas in (("MOV", "mov"), ("ADD", "add"), ("SUB", "sub"), ("CMP", "cmp"), ("SHR", "shr"), ("SHL", "shl"), ("SAR", "sar"),
                  ("XOR", "xor"), ("OR", "or"), ("AND", "and"), ("DEC", "dec"), ("INC", "inc"), ("TEST", "test"), ("LEA", "lea"),
                  ("IMUL", "imul"), ("BSF", "bsf"), ("ROL", "rol"), ("NEG", "neg"), ("BTS", "bts"), ("BSR", "bsr"), ("BSWAP", "bswap"),
                  ("ADC", "adc")):
    for suf, sz in (("B", 1), ("W", 2), ("L", 4), ("Q", 8)):
        OPS[base + suf] = (gas + suf.lower(), sz)
SSE = {"MOVOU": "movdqu", "MOVOA": "movdqa", "PXOR": "pxor", "MOVUPS": "movups"}
# BMI1 / BMI2 three-operand forms (zstd/seqdec_amd64.s, huff0/decompress_amd64.s): Plan 9 and AT&T write them in the same order
BMI3 = {"BZHIQ": "bzhiq", "BEXTRQ": "bextrq", "SHRXQ": "shrxq", "SHLXQ": "shlxq"}
CMOV = {"CMOVQEQ": "cmoveq", "CMOVQNE": "cmovneq"}
SETCC = {"SETGE": "setge"}
# sign-extending moves
SX = {"MOVWQSX": ("movswq", 2, 8)}
JCC = {"JMP": "jmp", "JEQ": "je", "JE": "je", "JNE": "jne", "JZ": "jz", "JNZ": "jnz", "JB": "jb", "JBE": "jbe", "JNA": "jna", "JA": "ja",
       "JAE": "jae", "JLE": "jle", "JG": "jg", "JGT": "jg", "JL": "jl", "JLT": "jl", "JGE": "jge", "JS": "js", "JNS": "jns", "JLS": "jbe", "JHI": "ja",
       "JCS": "jb", "JCC": "jae"}
# zero-extending moves: Plan 9 name -> (gas mnemonic, source width, destination width)
ZX = {"MOVBQZX": ("movzbq", 1, 8), "MOVBLZX": ("movzbl", 1, 4), "MOVWQZX": ("movzwq", 2, 8), "MOVWLZX": ("movzwl", 2, 4), "MOVLQZX": ("movl", 4, 4)}


class Fn:
    def __init__(self, name, frame, args):
        self.name, self.frame, self.args = name, frame, args
        self.lbase = 0                              # locals at 0(%rsp)
        self.abase = (frame + 15) & ~15             # arguments above them
        self.total = (self.abase + args + 15) & ~15


def split_operands(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


MEM = re.compile(r"^(?:(?P<sym>[A-Za-z_][A-Za-z0-9_]*)\+)?(?P<disp>[-+]?(?:0x[0-9a-fA-F]+|\d+))?\((?P<base>[A-Z0-9]+)\)(?:\((?P<idx>[A-Z0-9]+)\*(?P<scale>[1248])\))?$")
DATA = re.compile(r"^\xb7(?P<sym>\w+)\+(?P<disp>\d+)\(SB\)$")  # a package-level variable of the reference: provided by the C shim as p9data_<name>


DATA_SYMS = set()


def operand(fn, op, size):
    if op.startswith("$"):
        return op
    if re.fullmatch(r"X\d+", op):
        return "%xmm" + op[1:]
    if op in ("AH", "BH", "CH", "DH"):  # the high byte of AX..DX (MOVB AH, CL: the table-log byte of a decSymbol)
        if size != 1:
            raise ValueError("byte register in a wider instruction: " + op)
        return "%" + op.lower()
    if op in ("AL", "BL", "CL", "DL") or re.fullmatch(r"(SI|DI|BP|R\d+)B", op):  # explicit byte registers
        if size != 1:
            raise ValueError("byte register in a wider instruction: " + op)
        return "%" + REG8[op[0] + "X" if len(op) == 2 and op[1] == "L" else op[:-1]]
    if op in REG64:
        if size is None:
            raise ValueError("register width unknown: " + op)
        return "%" + BYSIZE[size][op]
    m = DATA.match(op)
    if m:
        DATA_SYMS.add(m.group("sym"))
        return "p9data_%s+%s(%%rip)" % (m.group("sym"), m.group("disp"))
    m = MEM.match(op)
    if not m:
        raise ValueError("operand not understood: " + op)
    disp = int(m.group("disp"), 0) if m.group("disp") else 0
    base = m.group("base")
    if base == "FP":
        if not m.group("sym") or m.group("idx"):
            raise ValueError("FP reference without a name: " + op)
        return "%d(%%rsp)" % (fn.abase + disp)
    if base == "SP":
        if m.group("sym") or m.group("idx"):
            raise ValueError("SP reference not understood: " + op)
        if disp < 0 or disp >= max(fn.frame, 1):
            raise ValueError("local outside the frame: " + op)
        return "%d(%%rsp)" % (fn.lbase + disp)
    if m.group("sym"):
        raise ValueError("symbol reference not understood: " + op)
    s = "%d" % disp if disp else ""
    s += "(%" + REG64[base]
    if m.group("idx"):
        s += ",%" + REG64[m.group("idx")] + "," + m.group("scale")
    return s + ")"


def expand_macros(lines):
    """The Go assembler's preprocessor, as far as the reference's files use it: `#define NAME text` and `#define name(a, b) body`
    with backslash continuations (a body's line breaks separate instructions), expanded until nothing changes."""
    obj, fun, out = {}, {}, []
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("#define"):
            body = ln
            while body.rstrip().endswith("\\"):
                i += 1
                body = body.rstrip()[:-1] + "\n" + lines[i]
            m = re.match(r"\s*#define\s+(\w+)\(([^)]*)\)\s*(.*)$", body, re.S)
            if m:
                fun[m.group(1)] = ([a.strip() for a in m.group(2).split(",") if a.strip()], m.group(3))
            else:
                m = re.match(r"\s*#define\s+(\w+)\s*(.*)$", body, re.S)
                if m and m.group(2).strip():
                    obj[m.group(1)] = m.group(2).split("//")[0].strip()
                else:
                    out.append(ln)  # a bare flag (#define GOAMD64_v3): left to the conditional logic below
            i += 1
            continue
        out.append(ln)
        i += 1
    if not obj and not fun:
        return out

    def expand(text):
        for _ in range(20):
            before = text
            for name, (params, body) in fun.items():
                def sub(m):
                    args = [a.strip() for a in m.group(1).split(",")] if m.group(1).strip() else []
                    b = body
                    for pn, av in zip(params, args):
                        b = re.sub(r"\b%s\b" % re.escape(pn), av, b)
                    return b
                text = re.sub(r"\b%s\(([^()]*)\)" % re.escape(name), sub, text)
            for name, val in obj.items():
                text = re.sub(r"(?<![\w\xb7])%s\b(?!\+)" % re.escape(name), val, text)
            if text == before:
                break
        return text
    res = []
    for ln in out:
        if ln.lstrip().startswith(("#", "//")) or ln.lstrip().startswith("TEXT"):
            res.append(ln)
            continue
        code = ln.split("//")[0]
        for piece in expand(code).split("\n"):
            # a label and an instruction may share a line after expansion ("loop: MOVQ ...")
            m = re.match(r"^(\s*\w+:)\s*(\S.*)$", piece)
            if m:
                res.append(m.group(1))
                res.append("\t" + m.group(2))
            else:
                res.append(piece)
    return res


def translate(src_lines):
    src_lines = expand_macros(src_lines)
    out = [".text"]
    fn = None
    skip = []  # preprocessor state: True while inside a dropped branch
    for ln, raw in enumerate(src_lines, 1):
        line = raw.split("//")[0].rstrip()
        if not line.strip():
            continue
        t = line.strip()
        if t.startswith("#"):
            if t.startswith("#include"):
                continue
            if t.startswith("#ifdef") or t.startswith("#ifndef"):
                # GOAMD64_v3 / v4 are never defined here: #ifdef X drops its branch, #ifndef X keeps it
                skip.append(t.startswith("#ifdef"))
            elif t.startswith("#else"):
                skip[-1] = not skip[-1]
            elif t.startswith("#endif"):
                skip.pop()
            elif t.startswith("#define"):
                pass
            else:
                raise ValueError("line %d: %s" % (ln, t))
            continue
        if any(skip):
            continue
        if t.startswith("TEXT"):
            m = re.match(r"TEXT\s+\xb7(\w+)\(SB\),\s*(?:[A-Z|]+,\s*)?\$(\d+)(?:-(\d+))?", t)
            if not m:
                raise ValueError("line %d: %s" % (ln, t))
            if fn is not None and fn != "skip":
                out.append("    ud2")
            if m.group(1).startswith(("cvtLZ4", "_dummy_")):  # the LZ4 converters are not on the encode path
                fn = "skip"
                continue
            fn = Fn(m.group(1), int(m.group(2)), int(m.group(3) or 0))
            n = fn.name
            out += ["", ".globl p9_%s" % n, ".type p9_%s, @function" % n, "p9_%s:" % n,
                    "    pushq %rbx", "    pushq %rbp", "    pushq %r12", "    pushq %r13", "    pushq %r14", "    pushq %r15", "    pushq %rdi",
                    "    subq $%d, %%rsp" % fn.total]
            for k in range(0, fn.args, 8):
                out += ["    movq %d(%%rdi), %%rax" % k, "    movq %%rax, %d(%%rsp)" % (fn.abase + k)]
            continue
        if fn == "skip":
            continue
        if fn is None:
            raise ValueError("line %d: code outside a function" % ln)
        if t.endswith(":"):
            out.append(".L%s_%s:" % (fn.name, t[:-1]))
            continue
        parts = t.split(None, 1)
        mn = parts[0]
        ops = split_operands(parts[1]) if len(parts) > 1 else []
        if mn == "RET":
            out.append("    movq %d(%%rsp), %%rdi" % fn.total)
            for k in range(0, fn.args, 8):
                out += ["    movq %d(%%rsp), %%rax" % (fn.abase + k), "    movq %%rax, %d(%%rdi)" % k]
            out += ["    addq $%d, %%rsp" % fn.total, "    popq %rdi", "    popq %r15", "    popq %r14", "    popq %r13", "    popq %r12", "    popq %rbp",
                    "    popq %rbx", "    ret"]
            continue
        if mn in JCC:
            out.append("    %s .L%s_%s" % (JCC[mn], fn.name, ops[0]))
            continue
        if mn in SSE:
            out.append("    %s %s" % (SSE[mn], ", ".join(operand(fn, o, None) for o in ops)))
            continue
        if mn in BMI3:
            out.append("    %s %s" % (BMI3[mn], ", ".join(operand(fn, o, 8) for o in ops)))
            continue
        if mn in CMOV:
            out.append("    %s %s, %s" % (CMOV[mn], operand(fn, ops[0], 8), operand(fn, ops[1], 8)))
            continue
        if mn in SETCC:
            out.append("    %s %s" % (SETCC[mn], operand(fn, ops[0], 1)))
            continue
        if mn in SX:
            g, sw, dw = SX[mn]
            out.append("    %s %s, %s" % (g, operand(fn, ops[0], sw), operand(fn, ops[1], dw)))
            continue
        if mn in ZX:
            g, sw, dw = ZX[mn]
            out.append("    %s %s, %s" % (g, operand(fn, ops[0], sw), operand(fn, ops[1], dw)))
            continue
        if mn == "CALL":  # only runtime.memmove(to, from, n), arguments at 0 / 8 / 16(SP) (Go's stack-based ABI0); every register is dead after it
            if ops != ["runtime\xb7memmove(SB)"]:
                raise ValueError("line %d: CALL %s" % (ln, ops))
            out += ["    movq %d(%%rsp), %%rdi" % fn.lbase, "    movq %d(%%rsp), %%rsi" % (fn.lbase + 8), "    movq %d(%%rsp), %%rdx" % (fn.lbase + 16),
                    "    call memmove@PLT"]
            continue
        if mn not in OPS:
            raise ValueError("line %d: mnemonic %s not handled" % (ln, mn))
        gas, size = OPS[mn]
        if mn.startswith("LEA"):
            o = [operand(fn, ops[0], 8), operand(fn, ops[1], size)]
        elif mn.startswith(("SHR", "SHL", "SAR", "ROL")) and (ops[0] in REG64 or ops[0] == "CL"):  # shift count in CX
            o = ["%cl", operand(fn, ops[1], size)]
        else:
            o = [operand(fn, x, size) for x in ops]
        if mn.startswith("CMP"):
            o.reverse()  # Plan 9 writes CMP the Intel way: CMPQ a, b sets the flags of a - b
        if mn == "MOVQ" and ops[0].startswith("$"):
            v = int(ops[0][1:], 0)
            if v > 0x7fffffff or v < -0x80000000:
                if not o[1].startswith("%"):
                    raise ValueError("line %d: 64-bit immediate to memory" % ln)
                gas = "movabsq"
        out.append("    %s %s" % (gas, ", ".join(o)))
    if fn is not None and fn != "skip":
        out.append("    ud2")
    for sym in sorted(DATA_SYMS):  # defined by the C shim in the same library: PC-relative access needs a non-preemptible symbol
        out.append(".hidden p9data_%s" % sym)
    out.append('.section .note.GNU-stack,"",@progbits')
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    with open(src, encoding="utf-8") as f:
        text = translate(f.read().splitlines())
    with open(dst, "w") as f:
        f.write(text)
