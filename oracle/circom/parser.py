"""Lexer + recursive-descent parser for the circom-2 subset the reference's circuits use.

Grammar follows the circom 2.1 language reference (operator precedence of circom's own grammar:
ternary < || < && < comparisons < | < ^ < & < shifts < + - < * / \\ % < ** < prefix).  Every node
keeps the (line, offset) of its first token: anonymous components are named
`<Template>_<line>_<offset>` by the compiler's syntax-sugar remover, which the interpreter mirrors.

AST (plain tuples):
  expr: ('num', v) ('ref', name, accesses) ('bin', op, a, b) ('un', op, a) ('tern', c, a, b)
        ('call', name, args) ('anon', name, args, inputs, line, off) ('arr', items)
        access = ('idx', expr) | ('dot', name)
  stmt: ('block', stmts) ('sig', kind, items, op, init, tuple_form) ('var', items) ('comp', items)
        ('assign', op, lhs, rhs) ('tassign', op, lhs_list, rhs) ('eqc', a, b) ('if', c, a, b)
        ('for', init, cond, step, body) ('while', c, body) ('return', e) ('assert', e)
        ('log', args) ('expr', e)
"""
import re

_TOKEN_RE = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0x[0-9a-fA-F]+|\d+)
  | (?P<id>[A-Za-z_$][A-Za-z0-9_$]*)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<op><==|==>|<--|-->|===|\*\*=|<<=|>>=|\*\*|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|\+=|-=|\*=|/=|\\=|%=|&=|\|=|\^=|[-+*/\\%&|^~!<>=?:;,.(){}\[\]])
""", re.X | re.S)


class ParseError(Exception):
    pass


def tokenize(text, fname="<src>"):
    toks = []
    pos = 0
    line = 1
    n = len(text)
    m_ = _TOKEN_RE.match
    while pos < n:
        m = m_(text, pos)
        if not m:
            raise ParseError(f"{fname}:{line}: unexpected character {text[pos]!r}")
        kind = m.lastgroup
        s = m.group()
        if kind != "ws":
            toks.append((kind, s, line, pos))
        line += s.count("\n")
        pos = m.end()
    toks.append(("eof", "", line, pos))
    return toks


_ASSIGN_OPS = {"=", "<==", "<--", "+=", "-=", "*=", "/=", "\\=", "%=", "**=", "<<=", ">>=", "&=", "|=", "^="}
_BIN_LEVELS = [
    ("||",),
    ("&&",),
    ("==", "!=", "<", ">", "<=", ">="),
    ("|",),
    ("^",),
    ("&",),
    ("<<", ">>"),
    ("+", "-"),
    ("*", "/", "\\", "%"),
]


class Parser:
    def __init__(self, text, fname="<src>"):
        self.fname = fname
        self.t = tokenize(text, fname)
        self.i = 0

    # ------------------------------------------------------------ token helpers
    def peek(self, k=0):
        return self.t[self.i + k]

    def at(self, s):
        return self.t[self.i][1] == s and self.t[self.i][0] in ("op", "id")

    def accept(self, s):
        if self.at(s):
            self.i += 1
            return True
        return False

    def expect(self, s):
        tk = self.t[self.i]
        if tk[1] != s:
            raise ParseError(f"{self.fname}:{tk[2]}: expected {s!r}, found {tk[1]!r}")
        self.i += 1
        return tk

    def ident(self):
        tk = self.t[self.i]
        if tk[0] != "id":
            raise ParseError(f"{self.fname}:{tk[2]}: expected identifier, found {tk[1]!r}")
        self.i += 1
        return tk[1]

    # ------------------------------------------------------------ file level
    def parse_file(self):
        """-> dict(includes=[...], functions={}, templates={}, main=None|(...))."""
        out = {"includes": [], "functions": {}, "templates": {}, "main": None}
        while self.peek()[0] != "eof":
            if self.accept("pragma"):
                while not self.accept(";"):
                    self.i += 1
            elif self.accept("include"):
                tk = self.peek()
                if tk[0] != "str":
                    raise ParseError(f"{self.fname}:{tk[2]}: include expects a string")
                self.i += 1
                self.accept(";")
                out["includes"].append(tk[1][1:-1])
            elif self.accept("function"):
                name = self.ident()
                params = self.param_list()
                body = self.block()
                out["functions"][name] = (params, body, self.fname)
            elif self.at("template"):
                self.i += 1
                while self.at("custom") or self.at("parallel"):
                    self.i += 1
                name = self.ident()
                params = self.param_list()
                body = self.block()
                out["templates"][name] = (params, body, self.fname)
            elif self.at("component"):
                self.i += 1
                self.expect("main")
                public = []
                if self.accept("{"):
                    self.expect("public")
                    self.expect("[")
                    while not self.at("]"):
                        public.append(self.ident())
                        self.accept(",")
                    self.expect("]")
                    self.expect("}")
                self.expect("=")
                call = self.expression()
                self.expect(";")
                if call[0] != "call":
                    raise ParseError(f"{self.fname}: main component must be a template call")
                out["main"] = (public, call[1], call[2])
            else:
                tk = self.peek()
                raise ParseError(f"{self.fname}:{tk[2]}: unexpected {tk[1]!r} at file level")
        return out

    def param_list(self):
        self.expect("(")
        ps = []
        while not self.at(")"):
            ps.append(self.ident())
            self.accept(",")
        self.expect(")")
        return ps

    # ------------------------------------------------------------ statements
    def block(self):
        self.expect("{")
        stmts = []
        while not self.at("}"):
            stmts.append(self.statement())
        self.expect("}")
        return ("block", stmts)

    def statement(self):
        tk = self.peek()
        s = tk[1]
        if tk[0] == "op":
            if s == "{":
                return self.block()
            if s == ";":
                self.i += 1
                return ("block", [])
        if tk[0] == "id":
            if s == "if":
                self.i += 1
                self.expect("(")
                c = self.expression()
                self.expect(")")
                a = self.statement()
                b = None
                if self.accept("else"):
                    b = self.statement()
                return ("if", c, a, b)
            if s == "for":
                self.i += 1
                self.expect("(")
                init = self.simple_statement()
                self.expect(";")
                cond = self.expression()
                self.expect(";")
                step = self.simple_statement()
                self.expect(")")
                body = self.statement()
                return ("for", init, cond, step, body)
            if s == "while":
                self.i += 1
                self.expect("(")
                c = self.expression()
                self.expect(")")
                return ("while", c, self.statement())
            if s == "return":
                self.i += 1
                e = self.expression()
                self.expect(";")
                return ("return", e)
            if s == "assert":
                self.i += 1
                self.expect("(")
                e = self.expression()
                self.expect(")")
                self.expect(";")
                return ("assert", e, tk[2])
            if s == "log":
                self.i += 1
                self.expect("(")
                args = []
                while not self.at(")"):
                    if self.peek()[0] == "str":
                        args.append(("str", self.peek()[1][1:-1]))
                        self.i += 1
                    else:
                        args.append(self.expression())
                    self.accept(",")
                self.expect(")")
                self.expect(";")
                return ("log", args)
        st = self.simple_statement()
        self.expect(";")
        return st

    def dims(self):
        ds = []
        while self.accept("["):
            ds.append(self.expression())
            self.expect("]")
        return ds

    def simple_statement(self):
        """declaration / substitution / constraint / expression statement, without the ';'."""
        tk = self.peek()
        s = tk[1]
        if tk[0] == "id":
            if s == "signal":
                self.i += 1
                kind = "mid"
                if self.accept("input"):
                    kind = "in"
                elif self.accept("output"):
                    kind = "out"
                if self.accept("{"):  # tags
                    while not self.accept("}"):
                        self.i += 1
                items = []
                tuple_form = False
                if self.accept("("):
                    tuple_form = True
                    while not self.at(")"):
                        items.append((self.ident(), self.dims()))
                        self.accept(",")
                    self.expect(")")
                else:
                    while True:
                        items.append((self.ident(), self.dims()))
                        if not self.accept(","):
                            break
                op = init = None
                if self.at("<==") or self.at("<--"):
                    op = self.peek()[1]
                    self.i += 1
                    init = self.expression()
                return ("sig", kind, items, op, init, tuple_form, tk[2])
            if s == "var":
                self.i += 1
                items = []
                if self.accept("("):
                    names = []
                    while not self.at(")"):
                        names.append((self.ident(), self.dims()))
                        self.accept(",")
                    self.expect(")")
                    init = None
                    if self.accept("="):
                        init = self.expression()
                    return ("vartuple", names, init)
                while True:
                    name = self.ident()
                    ds = self.dims()
                    init = None
                    if self.accept("="):
                        init = self.expression()
                    items.append((name, ds, init))
                    if not self.accept(","):
                        break
                return ("var", items)
            if s == "component":
                self.i += 1
                if self.at("parallel"):
                    self.i += 1
                items = []
                while True:
                    name = self.ident()
                    ds = self.dims()
                    init = None
                    if self.accept("="):
                        if self.at("parallel"):
                            self.i += 1
                        init = self.expression()
                    items.append((name, ds, init))
                    if not self.accept(","):
                        break
                return ("comp", items, tk[2])
        # tuple substitution: (a, b) <== expr
        if tk[0] == "op" and s == "(":
            save = self.i
            try:
                self.i += 1
                lhs = []
                while not self.at(")"):
                    if self.at("_"):
                        self.i += 1
                        lhs.append(None)
                    else:
                        lhs.append(self.postfix())
                    if not self.accept(","):
                        break
                self.expect(")")
                if self.at("<==") or self.at("<--") or self.at("="):
                    op = self.peek()[1]
                    self.i += 1
                    rhs = self.expression()
                    return ("tassign", op, lhs, rhs, tk[2])
            except ParseError:
                pass
            self.i = save
        e = self.expression()
        t2 = self.peek()
        s2 = t2[1]
        if t2[0] == "op":
            if s2 in _ASSIGN_OPS:
                self.i += 1
                rhs = self.expression()
                return ("assign", s2, e, rhs, tk[2])
            if s2 == "==>" or s2 == "-->":
                self.i += 1
                lhs = self.expression()
                return ("assign", "<==" if s2 == "==>" else "<--", lhs, e, tk[2])
            if s2 == "===":
                self.i += 1
                rhs = self.expression()
                return ("eqc", e, rhs, tk[2])
            if s2 == "++" or s2 == "--":
                self.i += 1
                return ("assign", "+=" if s2 == "++" else "-=", e, ("num", 1), tk[2])
        return ("expr", e, tk[2])

    # ------------------------------------------------------------ expressions
    def expression(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.expression()
            self.expect(":")
            b = self.expression()
            return ("tern", c, a, b)
        return c

    def binary(self, lvl):
        if lvl == len(_BIN_LEVELS):
            return self.power()
        ops = _BIN_LEVELS[lvl]
        a = self.binary(lvl + 1)
        while True:
            tk = self.peek()
            if tk[0] == "op" and tk[1] in ops:
                self.i += 1
                b = self.binary(lvl + 1)
                a = ("bin", tk[1], a, b)
            else:
                return a

    def power(self):
        a = self.prefix()
        while self.at("**"):
            self.i += 1
            b = self.prefix()
            a = ("bin", "**", a, b)
        return a

    def prefix(self):
        tk = self.peek()
        if tk[0] == "op" and tk[1] in ("-", "!", "~"):
            self.i += 1
            return ("un", tk[1], self.prefix())
        return self.postfix()

    def postfix(self):
        tk = self.peek()
        kind, s, line, off = tk
        if kind == "num":
            self.i += 1
            return ("num", int(s, 16) if s.startswith("0x") else int(s))
        if kind == "op":
            if s == "(":
                self.i += 1
                e = self.expression()
                self.expect(")")
                return e
            if s == "[":
                self.i += 1
                items = []
                while not self.at("]"):
                    items.append(self.expression())
                    self.accept(",")
                self.expect("]")
                return ("arr", items)
        if kind == "id":
            if s == "parallel":
                self.i += 1
                return self.postfix()
            self.i += 1
            if self.at("("):
                self.i += 1
                args = []
                while not self.at(")"):
                    args.append(self.expression())
                    self.accept(",")
                self.expect(")")
                if self.at("("):
                    # anonymous component: T(params)(inputs)
                    self.i += 1
                    inputs = []
                    while not self.at(")"):
                        inputs.append(self.expression())
                        self.accept(",")
                    self.expect(")")
                    return ("anon", s, args, inputs, line, off)
                return ("call", s, args)
            acc = []
            while True:
                if self.accept("["):
                    acc.append(("idx", self.expression()))
                    self.expect("]")
                elif self.at(".") and self.peek(1)[0] == "id":
                    self.i += 1
                    acc.append(("dot", self.ident()))
                else:
                    break
            return ("ref", s, acc)
        raise ParseError(f"{self.fname}:{line}: unexpected token {s!r} in expression")
