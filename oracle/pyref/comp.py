"""Component-tree bookkeeping for the literal Python oracle (test infrastructure).

A `Comp` mirrors one circom template instance.  Signals are declared in the
template's declaration order and classified as output / input / intermediate;
sub-components are recorded in creation order.  After evaluation the tree is
walked in circom's O0 numbering order ([EXT], unverified):

    index 0 = constant 1, then main's outputs, public inputs, private inputs,
    main's intermediates, then each sub-component's block depth-first in
    creation order; within a component: outputs, inputs, intermediates in
    declaration order.

Every signal element carries a *kept* flag that defines the compact witness
layout shipped by the product (DESIGN.md "layout kept-v1"):

    K  main input / output                         -> kept
    H  assigned by a hint  `<--`                    -> kept
    Q  assigned by `<==` with a quadratic RHS       -> kept
    L  assigned by `<==` with a linear RHS (alias,
       constant, linear combination)               -> dropped
    U  declared but never assigned (value 0)        -> kept (e.g. carry[k-1],
       lib/bigint.circom:78-93)

This is a proxy for circom's --O2 simplification (one signal eliminated per
linear constraint); the real compiler's choice of eliminated signal cannot be
reproduced without the compiler, hence "parity unpinned" for the ordering.
"""

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class AssertFailed(Exception):
    """Mirrors circom_runtime's 'Assert Failed' error (exception code 4)."""

    def __init__(self, where=""):
        super().__init__("Error: Assert Failed. " + where)


class Sig:
    __slots__ = ("name", "n", "v", "k", "kind", "scalar")

    def __init__(self, name, n, kind):
        self.name = name
        self.scalar = n is None
        self.n = 1 if n is None else n
        self.v = [0] * self.n
        self.k = ["U"] * self.n
        self.kind = kind

    # scalar helpers
    def set(self, val, how, i=0):
        self.v[i] = val % P
        self.k[i] = how
        return self.v[i]

    def setall(self, vals, how):
        assert len(vals) == self.n, (self.name, len(vals), self.n)
        self.v = [x % P for x in vals]
        self.k = [how] * self.n

    @property
    def val(self):
        return self.v[0]


class Comp:
    def __init__(self, template, is_main=False):
        self.template = template
        self.outs = []
        self.ins = []
        self.mids = []
        self.subs = []  # (name, Comp)
        self.is_main = is_main
        self.public = set()
        self.failed = []  # constraint failures recorded when soft=True

    # declarations -----------------------------------------------------
    def out(self, name, n=None):
        s = Sig(name, n, "out")
        self.outs.append(s)
        return s

    def inp(self, name, n=None):
        s = Sig(name, n, "in")
        self.ins.append(s)
        return s

    def mid(self, name, n=None):
        s = Sig(name, n, "mid")
        self.mids.append(s)
        return s

    def sub(self, name, comp):
        self.subs.append((name, comp))
        return comp

    # constraints --------------------------------------------------------
    def eq(self, a, b, where=""):
        """`a === b` (checked at witness time -> Assert Failed)."""
        if (a - b) % P != 0:
            raise AssertFailed(f"{self.template}: {where}")

    def check(self, cond, where=""):
        if not cond:
            raise AssertFailed(f"{self.template}: {where}")

    # walk ----------------------------------------------------------------
    def walk(self, prefix="main"):
        """Yield (full_name, value, kept_flag) in O0 order for this subtree
        (the leading constant-1 signal is NOT yielded here)."""
        if self.is_main:
            order = list(self.outs)
            order += [s for s in self.ins if s.name in self.public]
            order += [s for s in self.ins if s.name not in self.public]
            order += self.mids
        else:
            order = self.outs + self.ins + self.mids
        for s in order:
            main_io = self.is_main and s.kind in ("out", "in")
            if s.scalar:
                yield (f"{prefix}.{s.name}", s.v[0], "K" if main_io else s.k[0])
            else:
                for i in range(s.n):
                    yield (f"{prefix}.{s.name}[{i}]", s.v[i], "K" if main_io else s.k[i])
        for name, c in self.subs:
            yield from c.walk(f"{prefix}.{name}")


def is_kept(flag):
    return flag in ("K", "H", "Q", "U")


def witness_full(main):
    """O0-style witness: [1] + every declared signal."""
    return [1] + [v for _, v, _ in main.walk()]


def witness_kept(main):
    """Compact 'kept-v1' witness: [1] + kept signals in O0 order."""
    return [1] + [v for _, v, k in main.walk() if is_kept(k)]


def symbols_kept(main):
    """[(slot, name)] for the kept layout (slot 0 = 'one')."""
    out = [(0, "one")]
    slot = 1
    for name, _, k in main.walk():
        if is_kept(k):
            out.append((slot, name))
            slot += 1
    return out
