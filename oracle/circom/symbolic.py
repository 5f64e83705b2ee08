"""Constraint generation for the circom interpreter (TEST INFRASTRUCTURE).

`SymProgram` runs a circuit like `Program` but carries, next to every value, what the compiler's
constraint generator carries: a constant, a linear combination of signals, or one product of two
linear combinations plus a linear combination.  Every `<==` and `===` becomes one R1CS constraint
`A * B = C` over the O0 signal numbering -- the constraint system `circom --O0` writes to the `.r1cs`
file (no simplification: every alias, constant and linear definition is its own constraint), derived
from the reference's UNMODIFIED `.circom` sources.  `write_r1cs` / `write_sym` emit the iden3 binary
`.r1cs` and the `.sym` text of that system.

Uses: (1) an R1CS for `checkConstraints` that does not come from zkwg's own hand derivation
(zk-email-verify_amd/py/zkwg/r1cs.py); (2) the artefact pair from which the product derives the
signals the kept-v1 layout drops (include/zkwg.h zkwg_circuit_create_full).
"""
import os
import struct
from array import array

from .runtime import (Program, Inst, Sig, Frame, CircomError, AssertFailed, P, BINOPS, MASK, _Ctx,
                      _deepcopy, _flatten, _nest, iter_signals)


class L:
    """c + sum t[w] * signal_w  (t is never mutated after construction)."""
    __slots__ = ("v", "t", "c")

    def __init__(self, v, t, c):
        self.v = v
        self.t = t
        self.c = c


class Q:
    """A * B + C with A, B, C linear."""
    __slots__ = ("v", "A", "B", "C")

    def __init__(self, v, A, B, C):
        self.v = v
        self.A = A
        self.B = B
        self.C = C


class NQ:
    """a value that is not a quadratic expression of signals (only legal in hints / var code)."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = v


class _Pend:
    __slots__ = ("op", "v")

    def __init__(self, op, v):
        self.op = op
        self.v = v


_ZERO_T = {}


def val(x):
    return x if type(x) is int else x.v


def vals(x):
    if type(x) is list:
        return [vals(y) for y in x]
    if type(x) is tuple:
        return tuple(vals(y) for y in x)
    return x if type(x) is int else x.v


def _as_L(x):
    if type(x) is int:
        return L(x, _ZERO_T, x)
    return x


def _lscale(x, k):
    k %= P
    if k == 0:
        return 0
    if k == 1:
        return x
    return L(x.v * k % P, {w: c * k % P for w, c in x.t.items()}, x.c * k % P)


def _ladd(x, y):
    """L + L"""
    if len(x.t) < len(y.t):
        x, y = y, x
    t = dict(x.t)
    for w, c in y.t.items():
        n = (t.get(w, 0) + c) % P
        if n:
            t[w] = n
        else:
            t.pop(w, None)
    if not t:
        return (x.c + y.c) % P
    return L((x.v + y.v) % P, t, (x.c + y.c) % P)


def s_add(x, y):
    tx, ty = type(x), type(y)
    if tx is int:
        if ty is int:
            return (x + y) % P
        x, y, tx, ty = y, x, ty, tx
    # x symbolic
    if tx is L:
        if ty is int:
            return L((x.v + y) % P, x.t, (x.c + y) % P)
        if ty is L:
            return _ladd(x, y)
        if ty is Q:
            return Q((x.v + y.v) % P, y.A, y.B, _as_L(s_add(y.C, x)))
        return NQ((x.v + y.v) % P)
    if tx is Q:
        if ty is int or ty is L:
            return Q((x.v + val(y)) % P, x.A, x.B, _as_L(s_add(x.C, y)))
        return NQ((x.v + y.v) % P)
    return NQ((x.v + val(y)) % P)


def s_neg(x):
    tx = type(x)
    if tx is int:
        return (-x) % P
    if tx is L:
        return _lscale(x, P - 1)
    if tx is Q:
        return Q((-x.v) % P, _as_L(_lscale(x.A, P - 1)), x.B, _as_L(s_neg(x.C)))
    return NQ((-x.v) % P)


def s_sub(x, y):
    return s_add(x, s_neg(y))


def s_mul(x, y):
    tx, ty = type(x), type(y)
    if tx is int:
        if ty is int:
            return x * y % P
        x, y, tx, ty = y, x, ty, tx
    if ty is int:
        if tx is L:
            return _lscale(x, y)
        if tx is Q:
            k = y % P
            if k == 0:
                return 0
            return Q(x.v * k % P, _as_L(_lscale(x.A, k)), x.B, _as_L(s_mul(x.C, k)))
        return NQ(x.v * y % P)
    if tx is L and ty is L:
        return Q(x.v * y.v % P, x, y, L(0, _ZERO_T, 0))
    return NQ(x.v * y.v % P)


def s_div(x, y):
    if type(y) is int:
        if y == 0:
            raise CircomError("division by zero")
        return s_mul(x, pow(y, -1, P))
    vy = y.v
    if vy == 0:
        raise CircomError("division by zero")
    return NQ(val(x) * pow(vy, -1, P) % P)


class SymProgram(Program):
    """Program with constraint generation.  After `run`, `constraints()` / `write_r1cs()` are available."""

    def __init__(self, *a, **kw):
        self.coef_id = {}
        self.coefs = []
        self.c_ptr = array("Q", [0])  # one entry per linear combination (3 per constraint)
        self.c_w = array("I")
        self.c_k = array("I")
        self.n_constraints = 0
        self.root = None
        super().__init__(*a, **kw)

    # ------------------------------------------------------------------ constraint store
    def _put_lc(self, lc):
        if type(lc) is not int:
            t = lc.t
            c = lc.c
        else:
            t, c = _ZERO_T, lc % P
        ids = self.coef_id
        if c:
            k = ids.get(c)
            if k is None:
                k = ids[c] = len(self.coefs)
                self.coefs.append(c)
            self.c_w.append(0)
            self.c_k.append(k)
        for w, cf in t.items():
            k = ids.get(cf)
            if k is None:
                k = ids[cf] = len(self.coefs)
                self.coefs.append(cf)
            self.c_w.append(w)
            self.c_k.append(k)
        self.c_ptr.append(len(self.c_w))

    def emit(self, d, where):
        """constraint d = 0"""
        td = type(d)
        if td is int:
            return                      # tautology (or a failed constant check, reported by the value test)
        if td is L:
            if not d.t:
                return
            self._put_lc(0); self._put_lc(0); self._put_lc(d)
        elif td is Q:
            self._put_lc(d.A); self._put_lc(d.B); self._put_lc(s_neg(d.C))
        else:
            raise CircomError(f"non quadratic constraint at {where}")
        self.n_constraints += 1

    # ------------------------------------------------------------------ expressions
    def c_expr(self, n, ctx):
        k = n[0]
        if k == "_int":
            e = self.c_expr(n[1], ctx)
            return lambda f: val(e(f))
        if k == "bin":
            op = n[1]
            a = self.c_expr(n[2], ctx)
            b = self.c_expr(n[3], ctx)
            if op == "+":
                return lambda f: s_add(a(f), b(f))
            if op == "-":
                return lambda f: s_sub(a(f), b(f))
            if op == "*":
                return lambda f: s_mul(a(f), b(f))
            if op == "/":
                return lambda f: s_div(a(f), b(f))
            if op == "&&":
                return lambda f: 1 if (val(a(f)) != 0 and val(b(f)) != 0) else 0
            if op == "||":
                return lambda f: 1 if (val(a(f)) != 0 or val(b(f)) != 0) else 0
            fn = BINOPS[op]

            def other(f):
                x, y = a(f), b(f)
                r = fn(val(x), val(y))
                if type(x) is int and type(y) is int:
                    return r
                return NQ(r) if op in ("\\", "%", "**", "<<", ">>", "&", "|", "^") else r
            return other
        if k == "un":
            a = self.c_expr(n[2], ctx)
            if n[1] == "-":
                return lambda f: s_neg(a(f))
            if n[1] == "!":
                return lambda f: 1 if val(a(f)) == 0 else 0

            def compl(f):
                x = a(f)
                r = (val(x) ^ MASK) % P
                return r if type(x) is int else NQ(r)
            return compl
        if k == "tern":
            c = self.c_expr(n[1], ctx)
            a = self.c_expr(n[2], ctx)
            b = self.c_expr(n[3], ctx)
            return lambda f: a(f) if val(c(f)) != 0 else b(f)
        return super().c_expr(n, ctx)

    # control flow conditions and indices need plain ints
    def _c_int(self, n, ctx):
        e = self.c_expr(n, ctx)
        return lambda f: val(e(f))

    def s_if(self, n, ctx):
        c = self._c_int(n[1], ctx)
        a = self.c_stmt(n[2], ctx)
        b = self.c_stmt(n[3], ctx) if n[3] is not None else None
        if b is None:
            return lambda f: a(f) if c(f) != 0 else None
        return lambda f: a(f) if c(f) != 0 else b(f)

    def s_for(self, n, ctx):
        return super().s_for((n[0], n[1], ("_int", n[2]), n[3], n[4]), ctx)

    def s_while(self, n, ctx):
        return super().s_while((n[0], ("_int", n[1]), n[2]), ctx)

    def s_assert(self, n, ctx):
        return super().s_assert((n[0], ("_int", n[1]), n[2]), ctx)

    def s_assign(self, n, ctx):
        _, op, lhs, rhs, line = n
        if op not in ("=", "<==", "<--") and not (lhs[0] == "ref" and lhs[1] in ctx.comps and not ctx.is_function):
            # compound assignment on a variable: x op= e  ->  x = x op e with symbolic arithmetic
            return super().s_assign((n[0], "=", lhs, ("bin", op[:-1], lhs, rhs), line), ctx)
        return super().s_assign(n, ctx)

    # ------------------------------------------------------------------ signal access
    def sym_read(self, inst, name, idx):
        s = inst.sigs.get(name)
        if s is None:
            raise CircomError(f"{inst.name} ({inst.tname}): no signal {name}")
        v = Program.read_sig(self, inst, name, idx)
        nd, ni = len(s.dims), len(idx)
        flat = 0
        for i, st in zip(idx, s.strides):
            flat += i * st
        base = s.base + flat
        if ni == nd:
            return L(v, {base: 1}, 0)
        fl = _flatten(v, [])
        return _nest([L(x, {base + j: 1}, 0) for j, x in enumerate(fl)], s.dims[ni:])

    def read_sig(self, inst, name, idx):
        return self.sym_read(inst, name, idx)

    def c_ref_read(self, n, ctx):
        name, acc = n[1], n[2]
        prog = self
        if name in ctx.comps and not ctx.is_function:
            cidx, sname, sidx = self._split_comp_access(acc, ctx, name)
            if sname is None:
                raise CircomError(f"{ctx.fname}: component {name} used as a value")

            def rdc(f):
                inst = f.comps[name]
                for ci in cidx:
                    inst = inst[val(ci(f))]
                if inst is None:
                    raise CircomError(f"component {name} read before creation")
                if not inst.done:
                    prog.run_inst(inst)
                return prog.sym_read(inst, sname, tuple([val(i(f)) for i in sidx]))
            return rdc
        idx = [self.c_expr(a[1], ctx) for a in acc]
        if name in ctx.sigs and not ctx.is_function:
            return lambda f: prog.sym_read(f.inst, name, tuple([val(i(f)) for i in idx]))
        if not idx:
            def rv0(f):
                try:
                    return f.v[name]
                except KeyError:
                    raise CircomError(f"{ctx.fname}: {ctx.name}: undeclared symbol {name}") from None
            return rv0

        def rvn(f):
            v = f.v[name]
            for i in idx:
                v = v[val(i(f))]
            return v
        return rvn

    # ------------------------------------------------------------------ stores
    def _store(self, inst, s, idx, v, how):
        idx = tuple(val(i) for i in idx)
        Program._store(self, inst, s, idx, vals(v), how)
        if how == "<==" or how == "in<==":
            flat = 0
            for i, st in zip(idx, s.strides):
                flat += i * st
            base = s.base + flat
            where = f"{inst.name}.{s.name}"
            if type(v) is list or type(v) is tuple:
                for j, x in enumerate(_flatten(list(v) if type(v) is tuple else v, [])):
                    self.emit(s_sub(L(val(x), {base + j: 1}, 0), x), where)
            else:
                self.emit(s_sub(L(val(v), {base: 1}, 0), v), where)

    def _bind_input(self, inst, s):
        pend = inst.pending.pop(s.name, None)
        if pend is None:
            return
        for prefix, pv in pend.items():
            if type(pv) is _Pend:
                self._store(inst, s, prefix, pv.v, "in<==" if pv.op == "<==" else "in")
            else:
                self._store(inst, s, prefix, pv, "in")      # main inputs

    def c_anon(self, n, ctx):
        _, tname, args, inputs, line, off = n
        cargs = [self.c_expr(x, ctx) for x in args]
        cins = [self.c_expr(x, ctx) for x in inputs]
        cbase = f"{tname}_{line}_{off}"
        loop_id = ctx.loops[-1] if ctx.loops else None
        prog = self

        def run(f):
            t = prog.get_template(tname)
            cname = cbase if loop_id is None else f"{cbase}[{f.lc[loop_id]}]"
            inst = Inst(tname, [vals(a(f)) for a in cargs], cname)
            f.inst.subs.append(inst)
            if len(cins) != len(t.inputs):
                raise CircomError(f"{cname}: {len(cins)} inputs given, template declares {len(t.inputs)}")
            for nm, ci in zip(t.inputs, cins):
                inst.pending[nm] = {(): _Pend("<==", ci(f))}
            prog.run_inst(inst)
            outs = [prog.sym_read(inst, o, ()) for o in t.outputs]
            return outs[0] if len(outs) == 1 else tuple(outs)
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
            inst = Inst(tname, [vals(_deepcopy(a(f))) for a in cargs], cname)
            f.inst.subs.append(inst)
            return inst
        return mk

    def _c_store(self, lhs, op, ctx, line):
        name, acc = lhs[1], lhs[2]
        if not ctx.is_function and name in ctx.comps:
            cidx, sname, sidx = self._split_comp_access(acc, ctx, name)
            if sname is None:
                raise CircomError(f"{ctx.fname}:{line}: component assignment needs a template call")

            def stc(f, v):
                inst = f.comps[name]
                for ci in cidx:
                    inst = inst[val(ci(f))]
                if inst is None:
                    raise CircomError(f"{ctx.name}: component {name} used before creation")
                if inst.done or inst.running:
                    raise CircomError(f"{inst.name}.{sname} assigned after the component was executed")
                d = inst.pending.get(sname)
                if d is None:
                    d = inst.pending[sname] = {}
                key = tuple([val(i(f)) for i in sidx])
                if key in d:
                    raise CircomError(f"{inst.name}.{sname}{list(key)} assigned twice")
                d[key] = _Pend(op, v)
            return stc
        return super()._c_store(lhs, op, ctx, line)

    def s_eqc(self, n, ctx):
        a = self.c_expr(n[1], ctx)
        b = self.c_expr(n[2], ctx)
        where = f"=== at {os.path.basename(ctx.fname)}:{n[3]} in {ctx.name}"
        prog = self

        def run(f):
            x, y = a(f), b(f)
            if type(x) is list or type(y) is list:
                fx, fy = _flatten(x, []), _flatten(y, [])
                if len(fx) != len(fy):
                    raise CircomError(where + ": array sizes differ")
                pairs = zip(fx, fy)
            else:
                pairs = ((x, y),)
            for p, q in pairs:
                if val(p) != val(q):
                    prog._fail(where + f" [{f.inst.name}]")
                prog.emit(s_sub(p, q), where)
        return run

    # ------------------------------------------------------------------ run / export
    def run(self, inputs, main=None, public=None):
        self.root = super().run(inputs, main, public)
        return self.root

    def final_wire_map(self):
        """temp wire id -> O0 index (array); index 0 -> 0"""
        m = array("I", bytes(4 * self.next_wire))
        i = 1
        for _, _, s, j in iter_signals(self.root, with_names=False):
            m[s.base + j] = i
            i += 1
        return m, i

    def write_r1cs(self, path):
        """iden3 `.r1cs` (version 1): header, constraints, wire-to-label map."""
        root = self.root
        m, n_wires = self.final_wire_map()
        sigs = list(root.sigs.values())
        n_out = sum(s.size for s in sigs if s.kind == "out")
        n_pub = sum(s.size for s in sigs if s.kind == "in" and s.name in root.public)
        n_prv = sum(s.size for s in sigs if s.kind == "in" and s.name not in root.public)
        coef_b = [c.to_bytes(32, "little") for c in self.coefs]
        with open(path, "wb") as fh:
            fh.write(b"r1cs" + struct.pack("<II", 1, 3))
            hdr = struct.pack("<I", 32) + P.to_bytes(32, "little") + struct.pack("<IIIIQI", n_wires, n_out, n_pub, n_prv,
                                                                                  n_wires, self.n_constraints)
            fh.write(struct.pack("<IQ", 1, len(hdr)) + hdr)
            # constraint section size: sum over lcs of 4 + 36 * terms
            n_lc = len(self.c_ptr) - 1
            size = 4 * n_lc + 36 * len(self.c_w)
            fh.write(struct.pack("<IQ", 2, size))
            ptr, cw, ck = self.c_ptr, self.c_w, self.c_k
            buf = bytearray()
            pk = struct.Struct("<I").pack
            for i in range(n_lc):
                a, b = ptr[i], ptr[i + 1]
                # iden3 wants the terms of a linear combination sorted by wire id
                terms = sorted((m[cw[t]], ck[t]) for t in range(a, b))
                buf += pk(b - a)
                for w, k in terms:
                    buf += pk(w)
                    buf += coef_b[k]
                if len(buf) > (1 << 22):
                    fh.write(buf)
                    buf = bytearray()
            fh.write(buf)
            fh.write(struct.pack("<IQ", 3, 8 * n_wires))
            fh.write(array("Q", range(n_wires)).tobytes())
        return n_wires

    def constraints(self):
        """list of (A, B, C) dicts over final wire ids (small circuits / tests)."""
        m, _ = self.final_wire_map()
        out = []
        ptr, cw, ck = self.c_ptr, self.c_w, self.c_k
        for i in range(self.n_constraints):
            lcs = []
            for j in range(3):
                a, b = ptr[3 * i + j], ptr[3 * i + j + 1]
                lcs.append({m[cw[t]]: self.coefs[ck[t]] for t in range(a, b)})
            out.append(tuple(lcs))
        return out
