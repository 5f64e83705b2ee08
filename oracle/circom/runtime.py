"""Witness-level executor for parsed circom programs (test infrastructure, see package docstring).

Semantics restated from the circom 2.1 documentation:

* values live in GF(P); relational operators compare the signed lift (z > P//2 -> z - P); `\\` and
  `%` act on the [0,P) representatives; `/` multiplies by the inverse; `<<`, `>>`, `&`, `|`, `^`, `~`
  follow the compiler's 254-bit masked definitions; `var` arrays are zero-initialised and have value
  semantics;
* `<==` / `<--` assign a signal exactly once; `===`, `assert` are checked with the computed values
  and raise AssertFailed (circom_runtime's "Assert Failed", exception code 4);
* `component c = T(args)` creates the instance (this is also the moment the compiler links it into
  its parent: sub-components are numbered in creation order); the template body runs once the
  parent has supplied the inputs -- here on first read of one of its signals, or when the parent's
  body ends (values do not depend on that order: signals are single-assignment);
* anonymous components `T(args)(inputs)` are desugared like the compiler does: a component named
  `<T>_<line>_<offset>` of the call expression, inputs bound in the template's declaration order,
  the result being its output signal(s);
* signal numbering (O0): index 0 is the constant 1, then main's outputs, public inputs, private
  inputs and intermediates, then each sub-component's block depth first in creation order; inside
  a component: outputs, inputs, intermediates in declaration order (unverified
  against a real compiler -- none exists offline).
"""
import itertools
import os
import sys

from .parser import Parser, ParseError

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
HALF = P // 2
MASK = (1 << 254) - 1


class CircomError(Exception):
    pass


class AssertFailed(Exception):
    """circom_runtime: 'Error: Assert Failed.' (exception code 4)."""

    def __init__(self, where=""):
        super().__init__("Error: Assert Failed. " + where)


# --------------------------------------------------------------------------- field operators
def _val(z):
    return z - P if z > HALF else z


def _shl(a, b):
    if b <= HALF:
        return ((a << b) & MASK) % P if b < 254 else 0
    return _shr(a, P - b)


def _shr(a, b):
    if b <= HALF:
        return a >> b if b < 254 else 0
    return _shl(a, P - b)


def _div(a, b):
    if b == 0:
        raise CircomError("division by zero")
    return a * pow(b, -1, P) % P


def _idiv(a, b):
    if b == 0:
        raise CircomError("integer division by zero")
    return a // b


def _imod(a, b):
    if b == 0:
        raise CircomError("modulo by zero")
    return a % b


def _lt(a, b):
    if a <= HALF and b <= HALF:
        return 1 if a < b else 0
    return 1 if _val(a) < _val(b) else 0


def _le(a, b):
    if a <= HALF and b <= HALF:
        return 1 if a <= b else 0
    return 1 if _val(a) <= _val(b) else 0


BINOPS = {
    "+": lambda a, b: (a + b) % P,
    "-": lambda a, b: (a - b) % P,
    "*": lambda a, b: a * b % P,
    "/": _div,
    "\\": _idiv,
    "%": _imod,
    "**": lambda a, b: pow(a, b, P),
    "<<": _shl,
    ">>": _shr,
    "&": lambda a, b: (a & b) % P,
    "|": lambda a, b: (a | b) % P,
    "^": lambda a, b: (a ^ b) % P,
    "==": lambda a, b: 1 if a == b else 0,
    "!=": lambda a, b: 1 if a != b else 0,
    "<": _lt,
    ">": lambda a, b: _lt(b, a),
    "<=": _le,
    ">=": lambda a, b: _le(b, a),
    "&&": lambda a, b: 1 if (a != 0 and b != 0) else 0,
    "||": lambda a, b: 1 if (a != 0 or b != 0) else 0,
}


def _deepcopy(v):
    if type(v) is list:
        return [_deepcopy(x) for x in v]
    return v


def _flatten(v, out):
    if type(v) is list:
        for x in v:
            _flatten(x, out)
    else:
        out.append(v)
    return out


def _zeros(dims):
    if not dims:
        return 0
    if len(dims) == 1:
        return [0] * dims[0]
    return [_zeros(dims[1:]) for _ in range(dims[0])]


def _nest(flat, dims, start=0):
    """flat[start:...] -> nested list of shape dims."""
    if not dims:
        return flat[start]
    if len(dims) == 1:
        return flat[start:start + dims[0]]
    sub = 1
    for d in dims[1:]:
        sub *= d
    return [_nest(flat, dims[1:], start + i * sub) for i in range(dims[0])]


# --------------------------------------------------------------------------- runtime objects
class Sig:
    """One declared signal (array) of a template instance."""
    __slots__ = ("name", "kind", "dims", "strides", "size", "vals", "how", "base")

    def __init__(self, name, kind, dims):
        self.name = name
        self.kind = kind
        self.dims = dims
        st = []
        size = 1
        for d in reversed(dims):
            st.append(size)
            size *= d
        self.strides = st[::-1]
        self.size = size
        self.vals = [None] * size
        self.how = None  # optional per-element provenance ('<==' / '<--'), filled when tracking
        self.base = 0    # first temporary wire id of this signal (constraint generation, symbolic.py)


class Inst:
    """One template instance (a node of the component tree)."""
    __slots__ = ("tname", "args", "sigs", "subs", "pending", "done", "running", "name", "public")

    def __init__(self, tname, args, name):
        self.tname = tname
        self.args = args
        self.name = name
        self.sigs = {}      # declaration order (dict keeps insertion order)
        self.subs = []      # creation order
        self.pending = {}   # input name -> {index-prefix tuple: value}
        self.done = False
        self.running = False
        self.public = ()


class Frame:
    __slots__ = ("v", "inst", "comps", "lc")

    def __init__(self, v, inst=None):
        self.v = v
        self.inst = inst
        self.comps = {}
        self.lc = {}    # hidden per-loop iteration counters (the compiler's `anon_var_*`)


class _Ret:
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


class _Ctx:
    """Compile-time knowledge about one template / function body."""

    def __init__(self, prog, name, is_function, sigs, comps, fname):
        self.prog = prog
        self.name = name
        self.is_function = is_function
        self.sigs = sigs
        self.comps = comps
        self.fname = fname
        self.loops = []     # ids of the lexically enclosing loops while compiling
        self.nloops = 0


def _scan_decls(node, sigs, comps, order_inputs, order_outputs):
    k = node[0]
    if k == "block":
        for s in node[1]:
            _scan_decls(s, sigs, comps, order_inputs, order_outputs)
    elif k == "sig":
        for name, _ in node[2]:
            sigs.add(name)
            if node[1] == "in":
                order_inputs.append(name)
            elif node[1] == "out":
                order_outputs.append(name)
    elif k == "comp":
        for name, _, _ in node[1]:
            comps.add(name)
    elif k == "if":
        _scan_decls(node[2], sigs, comps, order_inputs, order_outputs)
        if node[3] is not None:
            _scan_decls(node[3], sigs, comps, order_inputs, order_outputs)
    elif k == "for":
        _scan_decls(node[1], sigs, comps, order_inputs, order_outputs)
        _scan_decls(node[4], sigs, comps, order_inputs, order_outputs)
    elif k == "while":
        _scan_decls(node[2], sigs, comps, order_inputs, order_outputs)


class Template:
    __slots__ = ("name", "params", "body", "inputs", "outputs", "fname")


class Program:
    """Parsed + compiled circom program.  `Program(main_file, include_paths)`; `run(inputs)`."""

    def __init__(self, main_file=None, include_paths=(), strict_unassigned=True, track_how=False):
        self.include_paths = list(include_paths)
        self.functions_src = {}
        self.templates_src = {}
        self.templates = {}
        self.functions = {}
        self.main = None
        self.loaded = set()
        self.strict = strict_unassigned
        self.track_how = track_how
        self.soft_failures = None   # list -> collect `===`/assert failures instead of raising
        self.logs = []
        self.anon_serial = 0
        self.next_wire = 1          # temporary wire ids in declaration order (0 = the constant 1)
        if main_file is not None:
            self.load(main_file)

    # ------------------------------------------------------------------ loading
    def _resolve(self, path, frm):
        cands = [os.path.join(os.path.dirname(frm), path)] if frm else [path]
        cands += [os.path.join(d, path) for d in self.include_paths]
        for c in cands:
            if os.path.isfile(c):
                return os.path.realpath(c)
        raise CircomError(f"include not found: {path!r} (from {frm}); searched {cands}")

    def load(self, path, frm=None):
        full = self._resolve(path, frm)
        if full in self.loaded:
            return
        self.loaded.add(full)
        with open(full) as fh:
            text = fh.read()
        self.load_text(text, full)

    def load_text(self, text, fname):
        try:
            ast = Parser(text, fname).parse_file()
        except ParseError as e:
            raise CircomError(str(e)) from None
        for inc in ast["includes"]:
            self.load(inc, fname)
        for k, v in ast["functions"].items():
            if k in self.functions_src or k in self.templates_src:
                raise CircomError(f"{fname}: symbol {k} declared twice")
            self.functions_src[k] = v
        for k, v in ast["templates"].items():
            if k in self.functions_src or k in self.templates_src:
                raise CircomError(f"{fname}: symbol {k} declared twice")
            self.templates_src[k] = v
        if ast["main"] is not None:
            self.main = ast["main"]

    # ------------------------------------------------------------------ compilation (lazy)
    def get_function(self, name):
        f = self.functions.get(name)
        if f is None:
            if name not in self.functions_src:
                raise CircomError(f"unknown function or template {name}")
            params, body, fname = self.functions_src[name]
            ctx = _Ctx(self, name, True, set(), set(), fname)
            holder = [params, None]
            self.functions[name] = holder       # allow recursion
            holder[1] = self.c_stmt(body, ctx)
            f = holder
        return f

    def get_template(self, name):
        t = self.templates.get(name)
        if t is None:
            if name not in self.templates_src:
                raise CircomError(f"unknown template {name}")
            params, body, fname = self.templates_src[name]
            sigs, comps, ins, outs = set(), set(), [], []
            _scan_decls(body, sigs, comps, ins, outs)
            ctx = _Ctx(self, name, False, sigs, comps, fname)
            t = Template()
            t.name, t.params, t.inputs, t.outputs, t.fname = name, params, ins, outs, fname
            self.templates[name] = t
            t.body = self.c_stmt(body, ctx)
        return t

    # ------------------------------------------------------------------ expressions
    def c_expr(self, n, ctx):
        k = n[0]
        if k == "num":
            v = n[1] % P
            return lambda f: v
        if k == "ref":
            return self.c_ref_read(n, ctx)
        if k == "bin":
            op = n[1]
            a = self.c_expr(n[2], ctx)
            b = self.c_expr(n[3], ctx)
            if op == "+":
                return lambda f: (a(f) + b(f)) % P
            if op == "-":
                return lambda f: (a(f) - b(f)) % P
            if op == "*":
                return lambda f: a(f) * b(f) % P
            if op == "&&":
                return lambda f: 1 if (a(f) != 0 and b(f) != 0) else 0
            if op == "||":
                return lambda f: 1 if (a(f) != 0 or b(f) != 0) else 0
            fn = BINOPS[op]
            return lambda f: fn(a(f), b(f))
        if k == "un":
            a = self.c_expr(n[2], ctx)
            if n[1] == "-":
                return lambda f: (-a(f)) % P
            if n[1] == "!":
                return lambda f: 1 if a(f) == 0 else 0
            return lambda f: (a(f) ^ MASK) % P
        if k == "tern":
            c = self.c_expr(n[1], ctx)
            a = self.c_expr(n[2], ctx)
            b = self.c_expr(n[3], ctx)
            return lambda f: a(f) if c(f) != 0 else b(f)
        if k == "arr":
            items = [self.c_expr(x, ctx) for x in n[1]]
            return lambda f: [it(f) for it in items]
        if k == "call":
            return self.c_call(n, ctx)
        if k == "anon":
            return self.c_anon(n, ctx)
        raise CircomError(f"cannot compile expression {k}")

    def c_call(self, n, ctx):
        name = n[1]
        args = [self.c_expr(x, ctx) for x in n[2]]
        if name in self.templates_src:
            raise CircomError(f"{ctx.fname}: template {name} used as a function "
                              "(only `component x = T(..)` and `T(..)(..)` are supported)")
        prog = self

        def call(f):
            params, body = prog.get_function(name)
            if len(params) != len(args):
                raise CircomError(f"{name}: expected {len(params)} arguments")
            fr = Frame({p: _deepcopy(a(f)) for p, a in zip(params, args)})
            r = body(fr)
            if r is None:
                raise CircomError(f"function {name} ended without return")
            return r.v
        return call

    def c_anon(self, n, ctx):
        _, tname, args, inputs, line, off = n
        if ctx.is_function:
            raise CircomError("anonymous component inside a function")
        cargs = [self.c_expr(x, ctx) for x in args]
        cins = [self.c_expr(x, ctx) for x in inputs]
        cbase = f"{tname}_{line}_{off}"
        loop_id = ctx.loops[-1] if ctx.loops else None
        prog = self

        def run(f):
            t = prog.get_template(tname)
            # inside a loop the compiler indexes the anonymous component with a hidden counter of
            # the innermost enclosing loop (syntax-sugar remover's `anon_var_<line>_<offset>`)
            cname = cbase if loop_id is None else f"{cbase}[{f.lc[loop_id]}]"
            inst = Inst(tname, [a(f) for a in cargs], cname)
            f.inst.subs.append(inst)
            if len(cins) != len(t.inputs):
                raise CircomError(f"{cname}: {len(cins)} inputs given, template declares {len(t.inputs)}")
            for nm, ci in zip(t.inputs, cins):
                inst.pending[nm] = {(): ci(f)}
            prog.run_inst(inst)
            outs = [prog.read_sig(inst, o, ()) for o in t.outputs]
            if len(outs) == 1:
                return outs[0]
            return tuple(outs)
        return run

    def read_sig(self, inst, name, idx):
        s = inst.sigs.get(name)
        if s is None:
            raise CircomError(f"{inst.name} ({inst.tname}): no signal {name}")
        nd = len(s.dims)
        ni = len(idx)
        if ni == nd:
            flat = 0
            for i, d, st in zip(idx, s.dims, s.strides):
                if i >= d:
                    raise CircomError(f"{inst.name}.{name}: index {i} out of bounds {d}")
                flat += i * st
            v = s.vals[flat]
            if v is None:
                if self.strict:
                    raise CircomError(f"{inst.name}.{name}{list(idx)} read before assignment")
                return 0
            return v
        if ni > nd:
            raise CircomError(f"{inst.name}.{name}: too many indices")
        flat = 0
        for i, d, st in zip(idx, s.dims, s.strides):
            if i >= d:
                raise CircomError(f"{inst.name}.{name}: index {i} out of bounds {d}")
            flat += i * st
        rest = s.dims[ni:]
        cnt = 1
        for d in rest:
            cnt *= d
        chunk = s.vals[flat:flat + cnt]
        if None in chunk:
            if self.strict:
                raise CircomError(f"{inst.name}.{name}{list(idx)}[..] read before assignment")
            chunk = [0 if x is None else x for x in chunk]
        return _nest(chunk, rest)

    def c_ref_read(self, n, ctx):
        name, acc = n[1], n[2]
        prog = self
        if name in ctx.comps and not ctx.is_function:
            cidx, sname, sidx = self._split_comp_access(acc, ctx, name)
            if sname is None:
                raise CircomError(f"{ctx.fname}: component {name} used as a value")
            read_sig = self.read_sig

            def rdc(f):
                inst = f.comps[name]
                for ci in cidx:
                    inst = inst[ci(f)]
                if inst is None:
                    raise CircomError(f"component {name} read before creation")
                if not inst.done:
                    prog.run_inst(inst)
                return read_sig(inst, sname, tuple([i(f) for i in sidx]))
            return rdc
        for a in acc:
            if a[0] != "idx":
                raise CircomError(f"{ctx.fname}: {name}: unexpected '.' access")
        idx = [self.c_expr(a[1], ctx) for a in acc]
        if name in ctx.sigs and not ctx.is_function:
            strict = self.strict
            if len(idx) == 0:
                def rs0(f):
                    s = f.inst.sigs[name]
                    if s.dims:
                        return prog.read_sig(f.inst, name, ())
                    v = s.vals[0]
                    if v is None:
                        if strict:
                            raise CircomError(f"{f.inst.name}.{name} read before assignment")
                        return 0
                    return v
                return rs0
            if len(idx) == 1:
                i0 = idx[0]

                def rs1(f):
                    s = f.inst.sigs[name]
                    if len(s.dims) != 1:
                        return prog.read_sig(f.inst, name, (i0(f),))
                    i = i0(f)
                    if i >= s.size:
                        raise CircomError(f"{f.inst.name}.{name}: index {i} out of bounds {s.size}")
                    v = s.vals[i]
                    if v is None:
                        if strict:
                            raise CircomError(f"{f.inst.name}.{name}[{i}] read before assignment")
                        return 0
                    return v
                return rs1
            return lambda f: prog.read_sig(f.inst, name, tuple([i(f) for i in idx]))
        # variable
        if len(idx) == 0:
            def rv0(f):
                try:
                    return f.v[name]
                except KeyError:
                    raise CircomError(f"{ctx.fname}: {ctx.name}: undeclared symbol {name}") from None
            return rv0
        if len(idx) == 1:
            i0 = idx[0]

            def rv1(f):
                try:
                    return f.v[name][i0(f)]
                except IndexError:
                    raise CircomError(f"{ctx.name}: {name}[{i0(f)}] out of bounds") from None
            return rv1

        def rvn(f):
            v = f.v[name]
            for i in idx:
                v = v[i(f)]
            return v
        return rvn

    def _split_comp_access(self, acc, ctx, name):
        cidx, sname, sidx = [], None, []
        for a in acc:
            if a[0] == "dot":
                if sname is not None:
                    raise CircomError(f"{ctx.fname}: {name}: nested '.' access is not supported")
                sname = a[1]
            elif sname is None:
                cidx.append(self.c_expr(a[1], ctx))
            else:
                sidx.append(self.c_expr(a[1], ctx))
        return cidx, sname, sidx

    # ------------------------------------------------------------------ statements
    def c_stmt(self, n, ctx):
        k = n[0]
        m = getattr(self, "s_" + k)
        return m(n, ctx)

    def s_block(self, n, ctx):
        stmts = [self.c_stmt(s, ctx) for s in n[1]]
        if len(stmts) == 1:
            return stmts[0]

        def run(f):
            for s in stmts:
                r = s(f)
                if r is not None:
                    return r
            return None
        return run

    def s_if(self, n, ctx):
        c = self.c_expr(n[1], ctx)
        a = self.c_stmt(n[2], ctx)
        b = self.c_stmt(n[3], ctx) if n[3] is not None else None
        if b is None:
            return lambda f: a(f) if c(f) != 0 else None
        return lambda f: a(f) if c(f) != 0 else b(f)

    def s_for(self, n, ctx):
        init = self.c_stmt(n[1], ctx)
        cond = self.c_expr(n[2], ctx)
        lid = ctx.nloops
        ctx.nloops += 1
        ctx.loops.append(lid)
        step = self.c_stmt(n[3], ctx)
        body = self.c_stmt(n[4], ctx)
        ctx.loops.pop()
        uses_counter = self._has_anon(n[4]) or self._has_anon(n[3])

        if not uses_counter:
            def run(f):
                init(f)
                while cond(f) != 0:
                    r = body(f)
                    if r is not None:
                        return r
                    step(f)
                return None
            return run

        def runc(f):
            init(f)
            f.lc[lid] = 0
            while cond(f) != 0:
                r = body(f)
                if r is not None:
                    return r
                step(f)
                f.lc[lid] += 1
            return None
        return runc

    def _has_anon(self, n):
        if type(n) is tuple:
            if n and n[0] == "anon":
                return True
            return any(self._has_anon(x) for x in n)
        if type(n) is list:
            return any(self._has_anon(x) for x in n)
        return False

    def s_while(self, n, ctx):
        cond = self.c_expr(n[1], ctx)
        lid = ctx.nloops
        ctx.nloops += 1
        ctx.loops.append(lid)
        body = self.c_stmt(n[2], ctx)
        ctx.loops.pop()

        def run(f):
            f.lc[lid] = 0
            while cond(f) != 0:
                r = body(f)
                if r is not None:
                    return r
                f.lc[lid] += 1
            return None
        return run

    def s_return(self, n, ctx):
        e = self.c_expr(n[1], ctx)
        return lambda f: _Ret(e(f))

    def _fail(self, where):
        if self.soft_failures is not None:
            self.soft_failures.append(where)
            return
        raise AssertFailed(where)

    def s_assert(self, n, ctx):
        e = self.c_expr(n[1], ctx)
        where = f"assert at {os.path.basename(ctx.fname)}:{n[2]} in {ctx.name}"
        prog = self

        def run(f):
            if e(f) == 0:
                prog._fail(where)
        return run

    def s_log(self, n, ctx):
        parts = [(None, x[1]) if x[0] == "str" else (self.c_expr(x, ctx), None) for x in n[1]]
        prog = self

        def run(f):
            prog.logs.append(" ".join(s if c is None else str(c(f)) for c, s in parts))
        return run

    def s_expr(self, n, ctx):
        e = self.c_expr(n[1], ctx)

        def run(f):
            e(f)
        return run

    def s_eqc(self, n, ctx):
        a = self.c_expr(n[1], ctx)
        b = self.c_expr(n[2], ctx)
        where = f"=== at {os.path.basename(ctx.fname)}:{n[3]} in {ctx.name}"
        prog = self

        def run(f):
            x, y = a(f), b(f)
            if x != y:
                if type(x) is list or type(y) is list:
                    fx, fy = _flatten(x, []), _flatten(y, [])
                    if fx == fy:
                        return
                prog._fail(where + f" [{f.inst.name}]")
        return run

    def s_var(self, n, ctx):
        items = []
        for name, dims, init in n[1]:
            cd = [self.c_expr(d, ctx) for d in dims]
            ci = self.c_expr(init, ctx) if init is not None else None
            items.append((name, cd, ci))

        def run(f):
            for name, cd, ci in items:
                if ci is not None:
                    v = ci(f)
                    f.v[name] = _deepcopy(v) if type(v) is list else v
                elif cd:
                    f.v[name] = _zeros([d(f) for d in cd])
                else:
                    f.v[name] = 0
        return run

    def s_vartuple(self, n, ctx):
        names = [nm for nm, _ in n[1]]
        ci = self.c_expr(n[2], ctx) if n[2] is not None else None

        def run(f):
            if ci is None:
                for nm in names:
                    f.v[nm] = 0
            else:
                vals = ci(f)
                for nm, v in zip(names, vals):
                    f.v[nm] = _deepcopy(v)
        return run

    def s_comp(self, n, ctx):
        if ctx.is_function:
            raise CircomError("component declared inside a function")
        items = []
        for name, dims, init in n[1]:
            cd = [self.c_expr(d, ctx) for d in dims]
            mk = None
            if init is not None:
                mk = self._c_instantiate(init, ctx)
            items.append((name, cd, mk))

        def none_arr(dims):
            if len(dims) == 1:
                return [None] * dims[0]
            return [none_arr(dims[1:]) for _ in range(dims[0])]

        def run(f):
            for name, cd, mk in items:
                if cd:
                    f.comps[name] = none_arr([d(f) for d in cd])
                elif mk is not None:
                    f.comps[name] = mk(f, name)
                else:
                    f.comps[name] = None
        return run

    def _c_instantiate(self, init, ctx):
        if init[0] != "call" or init[1] not in self.templates_src:
            raise CircomError(f"{ctx.fname}: component initialiser must be a template call, got {init[:2]}")
        tname = init[1]
        cargs = [self.c_expr(x, ctx) for x in init[2]]
        prog = self

        def mk(f, cname):
            t = prog.get_template(tname)
            if len(cargs) != len(t.params):
                raise CircomError(f"{tname}: expected {len(t.params)} parameters")
            inst = Inst(tname, [_deepcopy(a(f)) for a in cargs], cname)
            f.inst.subs.append(inst)
            return inst
        return mk

    def s_sig(self, n, ctx):
        if ctx.is_function:
            raise CircomError("signal declared inside a function")
        _, kind, items, op, init, tuple_form, line = n
        citems = [(name, [self.c_expr(d, ctx) for d in dims]) for name, dims in items]
        cinit = self.c_expr(init, ctx) if init is not None else None
        prog = self
        how = op

        def run(f):
            inst = f.inst
            made = []
            for name, cd in citems:
                dims = [d(f) for d in cd]
                if name in inst.sigs:
                    raise CircomError(f"{inst.name}: signal {name} declared twice")
                s = Sig(name, kind, dims)
                s.base = prog.next_wire
                prog.next_wire += s.size
                inst.sigs[name] = s
                made.append(s)
                if kind == "in":
                    prog._bind_input(inst, s)
            if cinit is not None:
                v = cinit(f)
                if tuple_form:
                    if type(v) is not tuple or len(v) != len(made):
                        raise CircomError(f"{inst.name}: tuple declaration arity mismatch")
                    for s, x in zip(made, v):
                        prog._store(inst, s, (), x, how)
                else:
                    if len(made) != 1:
                        raise CircomError("initialiser on a multi-signal declaration")
                    prog._store(inst, made[0], (), v, how)
        return run

    def _bind_input(self, inst, s):
        pend = inst.pending.pop(s.name, None)
        if pend is None:
            return
        for prefix, val in pend.items():
            self._store(inst, s, prefix, val, "in")

    def _store(self, inst, s, idx, v, how):
        """assign value (scalar or nested list) to s[idx...]"""
        nd = len(s.dims)
        ni = len(idx)
        flat = 0
        for i, d, st in zip(idx, s.dims, s.strides):
            if i >= d:
                raise CircomError(f"{inst.name}.{s.name}: index {i} out of bounds {d}")
            flat += i * st
        if ni == nd:
            if type(v) is list or type(v) is tuple:
                raise CircomError(f"{inst.name}.{s.name}: array assigned to a scalar signal")
            if s.vals[flat] is not None:
                raise CircomError(f"{inst.name}.{s.name}{list(idx)} assigned twice")
            s.vals[flat] = v % P
            if self.track_how:
                if s.how is None:
                    s.how = [None] * s.size
                s.how[flat] = how
            return
        if ni > nd:
            raise CircomError(f"{inst.name}.{s.name}: too many indices")
        vals = _flatten(v, [])
        cnt = 1
        for d in s.dims[ni:]:
            cnt *= d
        if len(vals) != cnt:
            raise CircomError(f"{inst.name}.{s.name}: assigning {len(vals)} values to {cnt} slots")
        for j, x in enumerate(vals):
            if s.vals[flat + j] is not None:
                raise CircomError(f"{inst.name}.{s.name}[{flat + j}] assigned twice")
            s.vals[flat + j] = x % P
        if self.track_how:
            if s.how is None:
                s.how = [None] * s.size
            for j in range(cnt):
                s.how[flat + j] = how

    def s_tassign(self, n, ctx):
        _, op, lhs, rhs, line = n
        cr = self.c_expr(rhs, ctx)
        stores = [self._c_store(l, op, ctx, line) if l is not None else None for l in lhs]

        def run(f):
            v = cr(f)
            if type(v) is not tuple or len(v) != len(stores):
                raise CircomError("tuple assignment arity mismatch")
            for st, x in zip(stores, v):
                if st is not None:
                    st(f, x)
        return run

    def _c_store(self, lhs, op, ctx, line):
        """-> fn(frame, value) storing into an lvalue (signal / component signal / var)."""
        if lhs[0] != "ref":
            raise CircomError(f"{ctx.fname}:{line}: invalid assignment target")
        name, acc = lhs[1], lhs[2]
        prog = self
        if not ctx.is_function and name in ctx.comps:
            cidx, sname, sidx = self._split_comp_access(acc, ctx, name)
            if sname is None:
                raise CircomError(f"{ctx.fname}:{line}: component assignment needs a template call")
            if op == "=":
                raise CircomError(f"{ctx.fname}:{line}: '=' used on a component signal")

            def stc(f, v):
                inst = f.comps[name]
                for ci in cidx:
                    inst = inst[ci(f)]
                if inst is None:
                    raise CircomError(f"{ctx.name}: component {name} used before creation")
                if inst.done or inst.running:
                    raise CircomError(f"{inst.name}.{sname} assigned after the component was executed")
                d = inst.pending.get(sname)
                if d is None:
                    d = inst.pending[sname] = {}
                key = tuple([i(f) for i in sidx])
                if key in d:
                    raise CircomError(f"{inst.name}.{sname}{list(key)} assigned twice")
                d[key] = v
            return stc
        for a in acc:
            if a[0] != "idx":
                raise CircomError(f"{ctx.fname}:{line}: {name}: unexpected '.' access")
        idx = [self.c_expr(a[1], ctx) for a in acc]
        if not ctx.is_function and name in ctx.sigs:
            if op == "=":
                raise CircomError(f"{ctx.fname}:{line}: '=' used on signal {name}")

            def sts(f, v):
                s = f.inst.sigs.get(name)
                if s is None:
                    raise CircomError(f"{f.inst.name}: signal {name} used before declaration")
                if s.kind == "in":
                    raise CircomError(f"{f.inst.name}: input signal {name} assigned inside its template")
                prog._store(f.inst, s, tuple([i(f) for i in idx]), v, op)
            return sts
        if op != "=":
            raise CircomError(f"{ctx.fname}:{line}: signal operator {op} used on variable {name}")
        if not idx:
            def stv0(f, v):
                f.v[name] = _deepcopy(v) if type(v) is list else v
            return stv0
        last = idx[-1]
        pre = idx[:-1]

        def stvn(f, v):
            a = f.v[name]
            for i in pre:
                a = a[i(f)]
            a[last(f)] = _deepcopy(v) if type(v) is list else v
        return stvn

    def s_assign(self, n, ctx):
        _, op, lhs, rhs, line = n
        if lhs[0] != "ref":
            raise CircomError(f"{ctx.fname}:{line}: invalid assignment target")
        name = lhs[1]
        # component creation: c = T(..) / c[i] = T(..)
        if not ctx.is_function and name in ctx.comps and op == "=" and all(a[0] == "idx" for a in lhs[2]):
            mk = self._c_instantiate(rhs, ctx)
            idx = [self.c_expr(a[1], ctx) for a in lhs[2]]

            def mkrun(f):
                if not idx:
                    f.comps[name] = mk(f, name)
                    return
                ii = [i(f) for i in idx]
                a = f.comps[name]
                for i in ii[:-1]:
                    a = a[i]
                if a[ii[-1]] is not None:
                    raise CircomError(f"{ctx.name}: component {name}{ii} created twice")
                a[ii[-1]] = mk(f, name + "".join(f"[{i}]" for i in ii))
            return mkrun
        if op in ("=", "<==", "<--"):
            st = self._c_store(lhs, op, ctx, line)
            cr = self.c_expr(rhs, ctx)

            def run(f):
                st(f, cr(f))
            return run
        # compound assignment on a variable
        bop = op[:-1]
        rd = self.c_expr(lhs, ctx)
        st = self._c_store(lhs, "=", ctx, line)
        cr = self.c_expr(rhs, ctx)
        if bop == "+":
            def runp(f):
                st(f, (rd(f) + cr(f)) % P)
            return runp
        fn = BINOPS[bop]

        def runc(f):
            st(f, fn(rd(f), cr(f)))
        return runc

    # ------------------------------------------------------------------ execution
    def run_inst(self, inst):
        if inst.done:
            return
        if inst.running:
            raise CircomError(f"{inst.name}: signal read while the component is still being wired "
                              "(an input is missing or the circuit has a combinational cycle)")
        inst.running = True
        t = self.get_template(inst.tname)
        if len(t.params) != len(inst.args):
            raise CircomError(f"{inst.tname}: expected {len(t.params)} parameters, got {len(inst.args)}")
        f = Frame(dict(zip(t.params, inst.args)), inst)
        r = t.body(f)
        if r is not None:
            raise CircomError(f"{inst.tname}: return inside a template")
        for child in inst.subs:
            if not child.done:
                self.run_inst(child)
        if inst.pending:
            raise CircomError(f"{inst.name} ({inst.tname}): input(s) {sorted(inst.pending)} assigned "
                              "but never declared")
        inst.running = False
        inst.done = True

    def run(self, inputs, main=None, public=None):
        """Execute the main component (file's `component main`, or main=(template, args)).
        `inputs`: dict name -> int | str | nested lists.  Returns the root Inst."""
        if main is None:
            if self.main is None:
                raise CircomError("no main component")
            public, tname, args = self.main
            ctx = _Ctx(self, "main", True, set(), set(), "<main>")
            fr = Frame({})
            args = [self.c_expr(a, ctx)(fr) for a in args]
        else:
            tname, args = main
            public = public or []
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 20000))
        try:
            root = Inst(tname, [a % P if type(a) is int else a for a in args], "main")
            root.public = tuple(public)

            def norm(v):
                if type(v) is list or type(v) is tuple:
                    return [norm(x) for x in v]
                return int(v) % P
            for k, v in inputs.items():
                root.pending[k] = {(): norm(v)}
            self.run_inst(root)
            self._check_assigned_inputs(root)
        finally:
            sys.setrecursionlimit(old)
        return root

    def _check_assigned_inputs(self, inst):
        for s in inst.sigs.values():
            if s.kind == "in" and None in s.vals:
                raise CircomError(f"Not all inputs have been set: {inst.name}.{s.name}")
        for c in inst.subs:
            self._check_assigned_inputs(c)


# --------------------------------------------------------------------------- O0 numbering / .sym
def _elem_names(s):
    if not s.dims:
        yield s.name
        return
    for idx in itertools.product(*[range(d) for d in s.dims]):
        yield s.name + "".join(f"[{i}]" for i in idx)


def iter_signals(root, with_names=True):
    """Yield (name, value, Sig, flat_index) in the O0 order (without the leading constant 1).
    Unassigned signals (e.g. lib/bigint.circom carry[k-1]) yield value 0."""
    stack = [(root, "main")]
    while stack:
        inst, path = stack.pop()
        sigs = list(inst.sigs.values())
        outs = [s for s in sigs if s.kind == "out"]
        ins = [s for s in sigs if s.kind == "in"]
        mids = [s for s in sigs if s.kind == "mid"]
        if inst.public:
            ins = [s for s in ins if s.name in inst.public] + [s for s in ins if s.name not in inst.public]
        for s in outs + ins + mids:
            if with_names:
                for j, nm in enumerate(_elem_names(s)):
                    v = s.vals[j]
                    yield path + "." + nm, (0 if v is None else v), s, j
            else:
                for j, v in enumerate(s.vals):
                    yield None, (0 if v is None else v), s, j
        for c in reversed(inst.subs):
            stack.append((c, path + "." + c.name))


def count_signals(root):
    n = 0
    stack = [root]
    while stack:
        inst = stack.pop()
        for s in inst.sigs.values():
            n += s.size
        stack.extend(inst.subs)
    return n


def write_sym(root, fh):
    """circom `.sym` text at O0: `labelIdx,witnessIdx,componentIdx,name` (labels == witness
    indices when nothing is simplified away; componentIdx is the DFS index of the owner)."""
    comp_ids = {}
    i = 1
    for name, _, s, _ in iter_signals(root):
        owner = name.rsplit(".", 1)[0]
        cid = comp_ids.setdefault(owner, len(comp_ids))
        fh.write(f"{i},{i},{cid},{name}\n")
        i += 1
    return i
