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


# Python scalars inside a kernel body (a module-level constant, a literal the rewrite leaves alone such as the operand of a unary
# minus folded by the parser): Numba types them like its literals
PY = {"pyint": "int64", "pyfloat": "float64", "pybool": "bool"}


def numba_binop(op, a, b):
    a, b = PY.get(a, a), PY.get(b, b)
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
# the same values.  Filled from the audit of the committed fixtures; an unlisted entry fails the test.  (hsr/ = the reference's
# handheld_super_resolution/ package.)
_ZERO = ("starts as the int literal 0 and is added to in float: Numba's unified float64 starts at 0.0, the same value, and every "
         "later operation is float64 in both")
_CH = "uint8 CFA entry or the int literal 0 (grey mode): Numba's unified int64 holds the same index"
_MAX0 = ("max(0, q) hands back the int 0 for q < 0 where Numba's unified float64 holds 0.0: the value only feeds q / power and "
         "exp(-0.5 q), float64 either way")
REVIEWED = {
    ("accumulate", "channel"): _CH,                                   # hsr/merge.py:349-354
    ("accumulate_ref", "channel"): _CH,                               # hsr/merge.py:191-194
    ("accumulate", "z"): _MAX0,                                       # hsr/merge.py:424
    ("accumulate_ref", "y"): _MAX0,                                   # hsr/merge.py:206-208
    ("compute_k", "D"): "clamp(x, 0, 1) hands back the ints 0 / 1 at the ends where Numba holds 0.0 / 1.0; D only enters "
                        "(1 - D) k1 + D k_denoise, whose value is the same (small integers are exact in float64)",  # hsr/kernels.py:219-227
    ("hard_threshold", "k1"): "1 / k_shrink (float64) or the int literal 1: Numba's unified float64 holds 1.0, the same value",  # hsr/kernels.py:230-237
    ("cuda_apply_noise_model", "d_sq_"): _ZERO,                       # hsr/robustness.py:515-529
    ("cuda_apply_noise_model", "sigma_sq_"): _ZERO,
    ("cuda_compute_guide_image", "g"): _ZERO,
    ("cuda_decimate_to_grey", "c"): _ZERO,                            # hsr/utils_image.py:353-357
    ("cuda_uspcale_dogson", "w_acc"): _ZERO,                          # hsr/robustness.py:397-418
    ("cuda_uspcale_dogson", "flow_x"): "int literal 0 (reference frame) or a float32 flow component: Numba's unified type is "
                                       "float64; the only use is (x + flow_x + 0.5) / s, float64 in both (int64 + float32 -> "
                                       "float64, and float32 -> float64 is exact)",  # hsr/robustness.py:371-384
    ("cuda_uspcale_dogson", "flow_y"): "like flow_x",
    ("cuda_compute_local_min", "mini"): "starts as the float literal +inf (float64), then min(mini, R[y, x]) hands back float32 "
                                        "values where Numba's unified float64 holds the same numbers; stored into the float32 "
                                        "map r",  # hsr/robustness.py:678-687
    ("denoise_power_median", "r"): "min(r_acc, max_frame_count) hands back the float64 sum or the int threshold where Numba "
                                   "holds the threshold as float64: radius_max (max_frame_count - r) / max_frame_count is a "
                                   "true division, float64 with the same value in both",  # hsr/utils_image.py:299-300
}

# max / min calls with a float32 AND a float64 argument: Python hands back one of the operands (possibly the float32 one),
# Numba the unified float64.  site ("function:line" of the reference's source) -> why the float32 result cannot continue in
# float32 arithmetic.  The sites of every call class come from a second, calls-only audit run (HHSR_REFSIM_AUDIT=calls,
# tests/golden/typing_audit_calls.json: minutes instead of the full audit's hours); an unlisted site fails the test.
MIXED_CALL_SITES = {
    "cuda_apply_noise_model:524": "sigma_sq_ += max(sigma_p_sq, sigma_t * sigma_t) (hsr/robustness.py:524): float32 local "
                                  "variance against the float64 curve value squared; the result is only ADDED to the float64 "
                                  "accumulator sigma_sq_ (int64 / float64 + float32 -> float64, float32 -> float64 exact)",
    "cuda_compute_local_min:685": "mini = min(mini, R[y, x]) (hsr/robustness.py:685): see REVIEWED[cuda_compute_local_min, mini]",
}


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


CALLS = os.path.join(HERE, "golden", "typing_audit_calls.json")


def test_max_min_abs_classes(audit):
    with open(CALLS) as f:
        sites = {(c[0], tuple(c[1]), c[2]): c[4] for c in json.load(f)["calls"]}
    bad = []
    for entry in audit["calls"]:
        name, args, res, n = entry[:4]
        if name == "abs":
            if res != args[0]:
                bad.append((name, args, res, n))
            continue
        margs = [PY.get(a, a) for a in args]
        if any(a not in INTS | FLOATS for a in margs):
            bad.append((name, args, res, n, "non-scalar argument"))
        if len({a for a in margs if a in FLOATS}) > 1:
            # float32 / float64 mix: Numba's result is float64, Python's may be the float32 operand — every site of the class
            # must have been reviewed
            where = sites.get((name, tuple(args), res))
            if where is None:
                bad.append((name, args, res, n, "float32 / float64 mix: class missing from the calls-only audit"))
            else:
                bad += [(name, args, res, n, f"float32 / float64 mix at the unreviewed site {w}") for w in where
                        if w not in MIXED_CALL_SITES]
    assert not bad, "\n".join(map(str, bad))
    # the two audits saw the same call classes (the calls-only run executes the same kernels on the same inputs)
    assert {(c[0], tuple(c[1]), c[2]) for c in audit["calls"]} == set(sites)


def test_local_variables_have_one_type_or_are_reviewed(audit):
    unreviewed = [(k, v, ts) for k, v, ts in audit["vars"] if (k, v) not in REVIEWED]
    assert not unreviewed, "\n".join(map(str, unreviewed))
    stale = [k for k in REVIEWED if list(k) not in [[k_, v_] for k_, v_, _ in audit["vars"]]]
    assert not stale, f"REVIEWED entries the audit no longer reports: {stale}"
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
    assert au.call("max", "k:1", np.int64(0), np.float64(-1.0)) == 0
    assert au.assign("k", "x", np.float32(1)) == np.float32(1)
    assert sum(au.ops.values()) == n0 + 3
    assert numba_binop("add", "int64", "float32") == "float64" and numba_binop("mul", "uint8", "float32") == "float32"
    assert numba_binop("pow", "float32", "int64") == "float32" and numba_binop("truediv", "int64", "int64") == "float64"
