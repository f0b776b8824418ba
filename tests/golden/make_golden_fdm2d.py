#!/usr/bin/env python3
"""Generates tests/golden/fdm2d_reference_cases.json: the dense K-bar / K-check matrices that the reference's own tests of
Fdm2d::get_matrices_sps expect (/root/reference/russell_pde/src/fdm_2d.rs, tests `get_matrices_work` and
`get_matrices_periodic_bcs_work`), together with the inputs of those tests.

Run in the authoring container only (it reads /root/reference):   python tests/golden/make_golden_fdm2d.py
What is written is DATA: grid sizes, coefficients, prescribed nodes and the expected matrix entries printed in the assertions.
"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/russell_pde/src/fdm_2d.rs"


def matrices_in(text):
    """every matrix literal of the form  "┌ ... ┐\\n\\  │ a b c │\\n\\ ... └ ... ┘"  as a list of rows"""
    out = []
    for m in re.finditer(r'"┌[^"]*?┘"', text, re.S):
        rows = []
        for line in m.group(0).split("\\n"):
            if "│" in line:
                body = line[line.index("│") + 1:line.rindex("│")]
                rows.append([float(t) for t in body.split()])
        out.append(rows)
    return out


def main():
    text = open(SRC, encoding="utf-8").read()
    t1 = text[text.index("fn get_matrices_work()"):text.index("fn get_matrices_periodic_bcs_work()")]
    t2 = text[text.index("fn get_matrices_periodic_bcs_work()"):text.index("fn get_vectors_works()")]
    m1, m2 = matrices_in(t1), matrices_in(t2)
    kbar, kcheck = m1[0], m1[1]
    assert len(kbar) == 9 and len(kbar[0]) == 9 and len(kcheck) == 9 and len(kcheck[0]) == 3
    # the Lagrange-multiplier form of the same test (get_matrices_lmm, fdm_2d.rs:1094-1131): constraints C (3 x 12), augmented M (15 x 15)
    cc, mm = m1[2], m1[3]
    assert len(cc) == 3 and len(cc[0]) == 12 and len(mm) == 15 and len(mm[0]) == 15
    kper, aper = m2[0], m2[1]
    assert len(kper) == 12 and len(kper[0]) == 12 and len(aper) == 12 and len(aper[0]) == 12
    cases = [
        {"name": "get_matrices_work", "cite": "russell_pde/src/fdm_2d.rs:1011-1090",
         "nx": 4, "ny": 3, "dx": 1.0, "dy": 1.0, "kx": 100.0, "ky": 300.0, "alpha": 0.0, "periodic_x": False, "periodic_y": False,
         "prescribed": [0, 4, 8], "nu": 9, "np": 3, "kk_bar_dense": kbar, "kk_check_dense": kcheck,
         "lmm_cc_dense": cc, "lmm_mm_dense": mm,
         "note": "the reference asserts the same dense matrices for Sym::No, YesLower, YesUpper and YesFull"},
        {"name": "get_matrices_periodic_bcs_work", "cite": "russell_pde/src/fdm_2d.rs:1134-1207",
         "nx": 3, "ny": 4, "dx": 1.0, "dy": 1.0, "kx": 1.0, "ky": 1.0, "alpha": 0.0, "periodic_x": True, "periodic_y": True,
         "prescribed": [], "nu": 12, "np": 0, "kk_bar_dense": kper, "kk_check_dense": None, "lmm_cc_dense": None, "lmm_mm_dense": aper},
    ]
    with open(os.path.join(HERE, "fdm2d_reference_cases.json"), "w") as fp:
        json.dump({"generated_by": "tests/golden/make_golden_fdm2d.py", "cases": cases}, fp, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
