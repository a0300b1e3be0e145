"""Audit of the typing emulation the golden fixtures rest on.

tests/golden/*.npz were produced by executing the reference's @cuda.jit kernel bodies as plain Python on NumPy scalars
(tools/refsim): NumPy-2 scalar promotion (NEP 50) stands in for Numba's type inference; no real Numba ran anywhere.
`HHSR_REFSIM_AUDIT=1 python -m tools.refsim.make_goldens` regenerates every fixture with each arithmetic operator, each
max / min / abs call and each assignment to a local name of the kernels routed through tools/refsim/audit.py, which records
the CLASS of the operation — (operator, operand types) -> result type, with counts — into tests/golden/typing_audit.json
(and checks that the audited run reproduces the committed fixtures bit for bit).  This test holds Numba's rules and fails on
any recorded class that Numba types differently from what the emulation produced, or that the table does not cover.

Numba's rules as encoded below (numba 0.5x sources; numba is not installed here, so the citations are by file / symbol):
  * scalar conversions — numba/core/typeconv/rules.py `_init_casting_rules`: promote int8 -> int16 -> int32 -> int64,
    float32 -> float64; SAFE uint8 -> int16, int16 -> float32, int32 -> float64, int64 -> float64 ("inconsistent with the
    above" in the source's own words), UNSAFE int32 -> float32 (hence int64 -> float32);
  * `+ - *`, `//`, `%` — numba/core/typing/builtins.py `BinOp` / `BinOpFloorDiv` / `BinOpMod`: cases
    integer_binop_cases + (float32, float32) -> float32 + (float64, float64) -> float64; the overload with the fewest
    unsafe conversions wins: int64 (+) float32 -> float64, uint8 / bool (+) float32 -> float32, float32 (+) float64 -> float64;
    two integers -> int64 for every pair that occurs here (one of them is always int64, an int literal or a range index);
  * `/` — `BinOpTrueDiv`: integers -> float64, otherwise as above;
  * `**` — `BinOpPower`: float32 ** integer stays float32 ("Ensure that float32 ** int doesn't go through DP
    computations") where NumPy gives float64: any occurrence with a float32 base fails this test (none occurs);
  * unary minus — `UnaryNegate`: type-preserving;
  * `abs` — type-preserving; `max` / `min` — `Max` / `Min` unify their arguments: the RESULT TYPE is the unified type while
    Python's builtins hand back one of the operands unchanged.  Harmless when the arguments have one type or mix an integer
    with a float (the value is the same and every later operator promotes alike); a float32 / float64 mix would let
    the emulation continue in float32 where Numba continues in float64: fails this test;
  * a local variable has ONE type per kernel specialisation, the unification of everything assigned to it
    (numba/core/typeinfer.py, `TypeVar.unify`): a variable that the emulation saw with more than one type is computed in
    the WIDER type by Numba also BEFORE its first wide assignment.  Every such variable must be listed in REVIEWED with the
    reason why that makes no difference.
math.* (tools/refsim/loader.py `_KMath`: sqrt / exp / modf type-preserving for float32, float64 otherwise; copysign float64;
floor / ceil int64) follows numba/cuda/mathdecl.py except that the CUDA target types math.floor / math.ceil as FLOAT
(`Math_unary`), not int64 like the CPU target — the kernels' only uses (hsr/merge.py:141-142 and the same lines of
`accumulate`) wrap the result in max(., 0) and int(): the same int64 either way.
"""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
AUDIT = os.path.join(HERE, "golden", "typing_audit.json")

INTS = {"bool", "uint8", "int8", "int16", "int32", "int64"}
FLOATS = {"float32", "float64"}
SMALL_INT = {"bool", "uint8", "int8", "int16"}  # safe to float32 (typeconv/rules.py)


def numba_float_mix(a, b):
    """Result of an arithmetic operator with at least one float operand."""
    if "float64" in (a, b):
        return "float64"
    if a == b == "float32":
        return "float32"
    other = b if a == "float32" else a
    assert other in INTS, (a, b)
    return "float32" if other in SMALL_INT else "float64"


def numba_binop(op, a, b):
    if a not in INTS | FLOATS or b not in INTS | FLOATS:
        return None  # not a scalar class Numba types here (arrays, tuples, Python scalars): must be in OTHER_OK
    if op in ("add", "sub", "mul", "floordiv", "mod"):
        if a in FLOATS or b in FLOATS:
            return numba_float_mix(a, b)
        return "int64"  # (every integer pair of the kernels has an int64 side)
    if op == "truediv":
        if a in FLOATS or b in FLOATS:
            return numba_float_mix(a, b)
        return "float64"
    if op == "pow":
        if a == "float32" and b in INTS:
            return "float32"  # BinOpPower: NOT what NumPy does
        if a in FLOATS or b in FLOATS:
            return numba_float_mix(a, b)
        return "int64"
    if op in ("lshift", "rshift", "and", "or", "xor"):
        return "int64" if a in INTS and b in INTS else None
    return None


# classes outside the scalar table that are understood: (op, lhs, rhs, result) -> why it is fine
OTHER_OK = {}

# local variables the emulation saw with more than one type: (kernel, variable) -> why Numba's single unified type computes
# the same values.  Filled from the audit of the committed fixtures; an unlisted entry fails the test.
REVIEWED = {}


@pytest.fixture(scope="module")
def audit():
    if not os.path.exists(AUDIT):
        pytest.skip("tests/golden/typing_audit.json not generated yet (HHSR_REFSIM_AUDIT=1 python -m tools.refsim.make_goldens)")
    with open(AUDIT) as f:
        return json.load(f)


def test_operator_classes_follow_numba(audit):
    bad, seen = [], 0
    for op, a, b, res, n in audit["ops"]:
        seen += 1
        if op == "neg":
            if res != a:
                bad.append((op, a, b, res, n, "unary minus must preserve the type"))
            continue
        want = numba_binop(op, a, b)
        if want is None:
            if (op, a, b, res) not in OTHER_OK:
                bad.append((op, a, b, res, n, "class not covered by the table"))
        elif want != res:
            bad.append((op, a, b, res, n, f"Numba types this {want}"))
    assert seen >= 10
    assert not bad, "\n".join(map(str, bad))


def test_max_min_abs_classes(audit):
    bad = []
    for name, args, res, n in audit["calls"]:
        if name == "abs":
            if res != args[0]:
                bad.append((name, args, res, n))
            continue
        fl = {a for a in args if a in FLOATS}
        if len(fl) > 1:  # float32 / float64 mix: Numba's result is float64, Python's may be the float32 operand
            bad.append((name, args, res, n, "float32 / float64 mix"))
        if any(a not in INTS | FLOATS for a in args):
            bad.append((name, args, res, n, "non-scalar argument"))
    assert not bad, "\n".join(map(str, bad))


def test_local_variables_have_one_type_or_are_reviewed(audit):
    unreviewed = [(k, v, ts) for k, v, ts in audit["vars"] if (k, v) not in REVIEWED]
    assert not unreviewed, "\n".join(map(str, unreviewed))
    assert audit["vars_single_type"] > 50


def test_audit_instrumentation_is_transparent():
    """tools.refsim.audit returns every value unchanged and records its class (no reference needed)."""
    import sys

    import numpy as np

    sys.path.insert(0, os.path.dirname(HERE))
    from tools.refsim import audit as au

    n0 = sum(au.ops.values())
    r = au.op("add", np.int64(2), np.float32(0.5))
    assert isinstance(r, np.float64) and r == 2.5
    assert au.op("imul", np.float32(3), np.float32(0.5)) == np.float32(1.5)
    assert au.neg(np.float32(1)).dtype == np.float32
    assert au.call("max", np.int64(0), np.float64(-1.0)) == 0
    assert au.assign("k", "x", np.float32(1)) == np.float32(1)
    assert sum(au.ops.values()) == n0 + 3
    assert numba_binop("add", "int64", "float32") == "float64" and numba_binop("mul", "uint8", "float32") == "float32"
    assert numba_binop("pow", "float32", "int64") == "float32" and numba_binop("truediv", "int64", "int64") == "float64"
