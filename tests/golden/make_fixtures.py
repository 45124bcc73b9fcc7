#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own test DATA.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box).  It reads the reference's test files AS TEXT and extracts numbers:

  * the `.sprs` matrix assets the reference's tests load (data files; the 8 distinct
    ones) are copied to tests/golden/assets/;
  * the inline inputs (`b`, dense matrices) and expected outputs (`x`) of the
    hot-path tests in tests/solver_tests.rs, and the exact-integer pins of
    tests/basic_tests.rs:1272-1336 (qr_1 / qr_2), are written to
    tests/golden/reference_cases.json together with the tolerance the reference
    test uses.

Nothing but numbers and test names is kept: no reference source text travels.
"""
import json
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# assets used by hot-path tests (duplicates in the reference collapse to one file)
ASSETS = {
    "cholsol_1.sprs": "cholsol_1.sprs",
    "cholsol_2.sprs": "cholsol_2.sprs",   # == qrsol_5.sprs
    "cholsol_5.sprs": "cholsol_5.sprs",   # == allsol_1.sprs == qrsol_4.sprs
    "cholsol_7.sprs": "cholsol_7.sprs",   # == allsol_2.sprs
    "lusol_3.sprs": "lusol_3.sprs",
    "lusol_4.sprs": "lusol_4.sprs",
    "lusol_6.sprs": "lusol_6.sprs",       # == qrsol_3.sprs
    "qrsol_8.sprs": "qrsol_8.sprs",       # == qrsol_9.sprs (200x201, not on the hot path; kept for completeness)
}
ALIAS = {"qrsol_5.sprs": "cholsol_2.sprs", "allsol_1.sprs": "cholsol_5.sprs", "qrsol_4.sprs": "cholsol_5.sprs",
         "allsol_2.sprs": "cholsol_7.sprs", "qrsol_3.sprs": "lusol_6.sprs", "qrsol_9.sprs": "qrsol_8.sprs"}

# test name -> how the reference calls the library and checks the result
#   call: function under test; args: (order, tol); check: "x" = compare with golden x (abs tol),
#   "residual" = |A x - b| <= tol elementwise, "error" = must return an error code,
#   "agree" = cholsol and lusol must agree within tol.
PLAN = {
    "lsolve_1":  dict(call="lsolve",  check="x", tol=1e-3, ref="tests/solver_tests.rs:4-49"),
    "ltsolve_1": dict(call="ltsolve", check="x", tol=1e-3, ref="tests/solver_tests.rs:52-88"),
    "usolve_1":  dict(call="usolve",  check="x", tol=1e-3, ref="tests/solver_tests.rs:91-127"),
    "utsolve_1": dict(call="utsolve", check="x", tol=1e-3, ref="tests/solver_tests.rs:130-166"),
    "lusol_1": dict(call="lusol", order=1, lutol=1e-12, check="x", tol=1e-3, ref="tests/solver_tests.rs:169-214"),
    "lusol_2": dict(call="lusol", order=1, lutol=1e-12, check="x", tol=1e-3, ref="tests/solver_tests.rs:217-256"),
    "lusol_3": dict(call="lusol", order=1, lutol=1e-12, check="x", tol=0.5, ref="tests/solver_tests.rs:259-352"),
    "lusol_4": dict(call="lusol", order=1, lutol=1e-12, check="x", tol=1e-5, ref="tests/solver_tests.rs:355-448"),
    "lusol_5": dict(call="lusol", order=1, lutol=1e-12, check="x", tol=1e-3, ref="tests/solver_tests.rs:451-487"),
    "lusol_6": dict(call="lusol", order=1, lutol=1e-12, check="x", tol=1e-5, ref="tests/solver_tests.rs:490-505"),
    "cholsol_1": dict(call="cholsol", order=0, check="x", tol=1e-3, ref="tests/solver_tests.rs:508-543"),
    "cholsol_2": dict(call="cholsol", order=0, check="x", tol=1e-6, ref="tests/solver_tests.rs:546-563"),
    "cholsol_3": dict(call="cholsol", order=0, check="x", tol=1e-3, ref="tests/solver_tests.rs:566-592"),
    "cholsol_4": dict(call="cholsol", order=0, check="x", tol=1e-3, ref="tests/solver_tests.rs:595-620"),
    "cholsol_5": dict(call="cholsol", order=0, check="x", tol=1e-6, ref="tests/solver_tests.rs:623-640"),
    "cholsol_6": dict(call="cholsol", order=0, check="error", expect="NotPositiveDefinite", ref="tests/solver_tests.rs:643-669"),
    "cholsol_7": dict(call="cholsol", order=0, check="residual", tol=1e-1, ref="tests/solver_tests.rs:672-684"),
    "allsol_1": dict(call="cholsol+lusol", order_chol=0, order_lu=1, lutol=1e-12, check="agree", tol=1e-12,
                     ref="tests/solver_tests.rs:928-946"),
    "allsol_2": dict(call="lusol", order=1, lutol=1e-12, check="residual", tol=1e-12, ref="tests/solver_tests.rs:949-968"),
    # qrsol: the consumer of usolve / utsolve.  check "x": golden x; "residual": |A x - b| <= res_tol
    # elementwise over the rows of A; "x+residual": both; transpose: the test transposes the loaded matrix first
    "qrsol_1": dict(call="qrsol", order=2, check="x", tol=1e-1, ref="tests/solver_tests.rs:687-711"),
    "qrsol_2": dict(call="qrsol", order=2, check="x", tol=1e-3, ref="tests/solver_tests.rs:714-753"),
    "qrsol_3": dict(call="qrsol", order=2, check="x", tol=1e-3, ref="tests/solver_tests.rs:756-771"),
    "qrsol_4": dict(call="qrsol", order=2, check="x", tol=1e-3, ref="tests/solver_tests.rs:774-789"),
    "qrsol_5": dict(call="qrsol", order=2, check="x", tol=1e-3, ref="tests/solver_tests.rs:792-807"),
    "qrsol_6": dict(call="qrsol", order=2, check="x", tol=1e-3, ref="tests/solver_tests.rs:810-846"),
    "qrsol_7": dict(call="qrsol", order=2, check="x+residual", tol=1e-3, res_tol=1e-6, ref="tests/solver_tests.rs:849-890"),
    "qrsol_8": dict(call="qrsol", order=2, check="residual", res_tol=1e-6, ref="tests/solver_tests.rs:893-906"),
    "qrsol_9": dict(call="qrsol", order=2, check="residual", res_tol=1e-1, transpose=True, ref="tests/solver_tests.rs:909-925"),
    "allsol_1_qr": dict(test="allsol_1", call="qrsol+cholsol", order=2, order_chol=0, check="agree", tol=1e-12,
                        ref="tests/solver_tests.rs:928-946"),
    "allsol_2_qr": dict(test="allsol_2", call="qrsol+lusol", order=2, order_lu=1, lutol=1e-12, check="agree+residual", tol=1e-1,
                        res_tol=1e-12, ref="tests/solver_tests.rs:949-968"),
}


def split_tests(text):
    out = {}
    for chunk in re.split(r"\n#\[test\]\n", text)[1:]:
        m = re.search(r"fn (\w+)\(\)", chunk)
        if m:
            out[m.group(1)] = chunk
    return out


def strip_comments(s):
    return re.sub(r"//[^\n]*", "", s)


def parse_vec(body):
    """body = text after `vec!` starting at '[' ; returns (python list, end index)."""
    assert body[0] == "["
    depth, i = 0, 0
    for i, ch in enumerate(body):
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
            if depth == 0:
                break
    inner = body[: i + 1].replace("vec!", "")
    inner = re.sub(r",\s*\]", "]", inner)          # trailing commas
    inner = re.sub(r"(?<![\w.])(\d+)\.(?!\d)", r"\1.0", inner)  # `0.` -> `0.0`
    return json.loads(inner), i + 1


def let_vectors(chunk):
    chunk = strip_comments(chunk)
    vals = {}
    for m in re.finditer(r"let (?:mut )?(\w+)\s*=\s*vec!\s*\[", chunk):
        name = m.group(1)
        v, _ = parse_vec(chunk[m.end() - 1:])
        vals[name] = v
    return vals


def main():
    os.makedirs(os.path.join(HERE, "assets"), exist_ok=True)
    for dst, src in ASSETS.items():
        shutil.copyfile(os.path.join(REF, "tests/assets", src), os.path.join(HERE, "assets", dst))

    cases = {}
    tests = split_tests(open(os.path.join(REF, "tests/solver_tests.rs")).read())
    for name, plan in PLAN.items():
        chunk = tests[plan.get("test", name)]
        vals = let_vectors(chunk)
        case = {k: v for k, v in plan.items() if k != "test"}
        m = re.search(r'load\("\.?/?tests/assets/(\w+\.sprs)"\)', chunk)
        if m:
            a = m.group(1)
            case["asset"] = ALIAS.get(a, a)
        else:
            dense = [v for k, v in vals.items() if v and isinstance(v[0], list)]
            assert len(dense) == 1, name
            case["dense"] = dense[0]
        flat = {k: v for k, v in vals.items() if not (v and isinstance(v[0], list))}
        for key in ("b", "b_static", "b_chol"):
            if key in flat:
                case["b"] = flat[key]
                break
        if plan["check"] in ("x", "x+residual"):
            case["x"] = flat["x"]
        cases[name] = case

    # README / crate-doc known answer (README.md:132-182 == src/lib.rs:125-175): lusol(order=1, tol=1e-6), full precision
    readme = open(os.path.join(REF, "README.md")).read()
    m = re.search(r"\[(0\.2646806068156303[^\]]*)\]", readme)
    x_readme = [float(t) for t in m.group(1).split(",")]
    cases["readme_lusol"] = dict(call="lusol", order=1, lutol=1e-6, check="x_bitwise", tol=0.0,
                                 dense=cases["lusol_2"]["dense"], b=cases["lusol_2"]["b"], x=x_readme,
                                 ref="README.md:132-182")

    # exact-integer structure pins: tests/basic_tests.rs qr_1 / qr_2
    btests = split_tests(open(os.path.join(REF, "tests/basic_tests.rs")).read())
    pins = {}
    for name, asset in (("qr_1", "cholsol_2.sprs"), ("qr_2", "cholsol_1.sprs")):
        chunk = strip_comments(btests[name])
        pin = dict(asset=asset, order=2, ref="tests/basic_tests.rs:1272-1336")
        for m in re.finditer(r"assert_eq!\(\s*([\w.]+?)(?:\.unwrap\(\))?\s*,\s*(vec!\s*\[|\d+)", chunk):
            key = m.group(1)
            if m.group(2).startswith("vec"):
                v, _ = parse_vec(chunk[m.end() - 1:])
            else:
                v = int(m.group(2))
            pin[key] = v
        for m in re.finditer(r"assert_eq_f_vec\(&([\w.]+),\s*&vec!\s*\[", chunk):
            v, _ = parse_vec(chunk[m.end() - 1:])
            pin[m.group(1) + "~1e-6"] = v
        pins[name] = pin

    with open(os.path.join(HERE, "reference_cases.json"), "w") as f:
        json.dump(dict(solver=cases, structure=pins), f)
    print("wrote", len(cases), "solver cases,", len(pins), "structure pins")
    for k, v in pins.items():
        print(k, sorted(v.keys()))


if __name__ == "__main__":
    main()
