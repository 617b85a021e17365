#!/usr/bin/env python3
"""Writes tests/golden/kat_cases.json: hand-computable known-answer cases for the
density hot path.  Every expected value below is typed in by hand from the
definitions in SURVEY.md Appendix B (reference semantics:
density_clustering.cpp:126-195 pops, :197-212 FE, :230-288 NN) -- nothing here is
computed by the oracle or by the HIP path, so the file pins both.

Conventions: "none" neighbour = index n_rows+1, d2 = FLT_MAX (density_clustering.cpp:242-245).
"""
import json
import os

FLT_MAX = 3.4028234663852886e38

CASES = [
    {
        "name": "line_1d_strict_less",
        "doc": "x=0,1,2,4 on a line. r=1: d2=1 is NOT < 1 (strict '<', :178). r=1.5: rad2=2.25.",
        "coords": [[0.0], [1.0], [2.0], [4.0]],
        "radii": [1.0, 1.5],
        "pops": [[1, 1, 1, 1], [2, 3, 2, 1]],
        "fe_from_radius": 1.5,
        # nn: frame1 is equidistant to 0 and 2 -> lowest index wins (scan j ascending, strict '<', :270)
        "nn_idx": [1, 0, 1, 2],
        "nn_d2": [1.0, 1.0, 1.0, 4.0],
        # pops 2,3,2,1 -> fe strictly decreasing in pop. lower-FE sets: f0:{1} f1:{} f2:{1} f3:{0,1,2}
        "hd_idx": [1, 5, 1, 2],
        "hd_d2": [1.0, FLT_MAX, 1.0, 4.0],
    },
    {
        "name": "duplicates_are_neighbours",
        "doc": "two identical frames + one at (3,4). CPU path accepts d2=0 neighbours (:262 only skips i==j). r=5: 25 < 25 false.",
        "coords": [[0.0, 0.0], [0.0, 0.0], [3.0, 4.0]],
        "radii": [5.0, 5.5],
        "pops": [[2, 2, 1], [3, 3, 3]],
        "fe_from_radius": 5.0,
        "nn_idx": [1, 0, 0],
        "nn_d2": [0.0, 0.0, 25.0],
        # pops 2,2,1: frames 0,1 tie at the minimum FE -> no lower-FE frame; frame 2 -> {0,1}, both d2=25 -> j=0
        "hd_idx": [4, 4, 0],
        "hd_d2": [FLT_MAX, FLT_MAX, 25.0],
    },
    {
        "name": "all_equal_pops_no_hd",
        "doc": "two isolated frames: pops all 1, FE all -0.0, nobody has a lower-FE neighbour.",
        "coords": [[0.0], [10.0]],
        "radii": [1.0],
        "pops": [[1, 1]],
        "fe_from_radius": 1.0,
        "nn_idx": [1, 0],
        "nn_d2": [100.0, 100.0],
        "hd_idx": [3, 3],
        "hd_d2": [FLT_MAX, FLT_MAX],
    },
    {
        "name": "grid_2d_square",
        "doc": "unit square corners + centre. d2 corner-corner = 1 or 2, corner-centre = 0.5. r=0.75 (rad2=0.5625): centre sees 4, corners see centre. r=1.25 (1.5625): corners also see 2 adjacent corners.",
        "coords": [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [0.5, 0.5]],
        "radii": [0.75, 1.25],
        "pops": [[2, 2, 2, 2, 5], [4, 4, 4, 4, 5]],
        "fe_from_radius": 0.75,
        "nn_idx": [4, 4, 4, 4, 0],
        "nn_d2": [0.5, 0.5, 0.5, 0.5, 0.5],
        "hd_idx": [4, 4, 4, 4, 6],
        "hd_d2": [0.5, 0.5, 0.5, 0.5, FLT_MAX],
    },
    {
        "name": "dims_7_tail",
        "doc": "D=7 exercises the 4-lane body + pair tail + scalar tail; all values exact in binary so any order gives the same d2. frames: origin, e1*2, ones*1 (d2=7), ones*-1.",
        "coords": [[0, 0, 0, 0, 0, 0, 0], [2, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1],
                   [-1, -1, -1, -1, -1, -1, -1]],
        "radii": [2.5, 3.0],
        # d2: (0,1)=4 (0,2)=7 (0,3)=7 (1,2)=1+6=7 (1,3)=9+6=15 (2,3)=28 ; rad2 = 6.25 / 9
        "pops": [[2, 2, 1, 1], [4, 3, 3, 2]],
        "fe_from_radius": 3.0,
        "nn_idx": [1, 0, 0, 0],
        "nn_d2": [4.0, 4.0, 7.0, 7.0],
        # pops 4,3,3,2 -> f0: none; f1: {0}; f2: {0}; f3: {0,1,2} nearest 0 (7)
        "hd_idx": [5, 0, 0, 0],
        "hd_d2": [FLT_MAX, 4.0, 7.0, 7.0],
    },
]


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_cases.json")
    with open(out, "w") as f:
        json.dump({"flt_max": FLT_MAX, "cases": CASES}, f, indent=1)
    print("wrote", out, len(CASES), "cases")


if __name__ == "__main__":
    main()
