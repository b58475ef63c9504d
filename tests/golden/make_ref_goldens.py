#!/usr/bin/env python3
"""Generate tests/golden/ref_cases/*.{txt,json}: golden vectors from the PARTIAL reference build.

Run in the build container only (needs /root/reference and g++):

    make -C oracle/ref_partial && python tests/golden/make_ref_goldens.py [out_dir]

(tests/test_oracle.py::test_committed_goldens_regenerate re-runs this into a temporary directory and compares the files
byte for byte whenever the reference tree is present, so the fixtures cannot drift from their generator again.)

oracle/_ref/ref_driver links the reference's own header-only scoring code (src/methods.h score_permute /
merge_scores, src/gcre_paths.h PathSet, src/gcre.h need_flip, src/gcre_types.h Score) and its text-dump parser
(test/test.cpp), compiled from /root/reference where they lie; the JoinExec members of src/join_base.cpp (which
needs <Rcpp.h>) are the driver's own restatement -- see oracle/ref_partial/ref_driver.cpp.  The files written here
are data only: the input dump of every case and the outputs the reference code printed for it.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from geneticscre_amd import api  # noqa: E402
from geneticscre_amd.harness_io import problem_digest, write_problem, write_problem_bin  # noqa: E402
from geneticscre_amd.synth import make_problem  # noqa: E402
from helpers import SLOW_WIDE_CASES, WIDE_CASES, small_table, wide_problem  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
OUT = os.path.join(HERE, "ref_cases")

CASES = [
    # name, method, genes, edges, cases, ctrls, perm rows, iterations, length, top_k, seed, table
    ("m1_len5_ragged", "method1", 22, 55, 37, 33, 29, 29, 5, 9, 101, "small"),
    ("m2_len5_ragged", "method2", 22, 55, 37, 33, 29, 29, 5, 9, 102, "small"),
    ("m1_sentinel_k0", "method1", 9, 14, 10, 12, 0, 0, 3, 400, 103, "small"),
    ("m2_three_words_hyper", "method2", 18, 40, 70, 60, 21, 21, 4, 12, 104, "hyper"),
    ("m1_perm_rows_reused", "method1", 16, 36, 20, 25, 5, 13, 4, 6, 105, "small"),
    ("m2_perm_rows_truncated", "method2", 16, 36, 20, 25, 17, 8, 4, 6, 106, "hyper"),
    ("m1_all_ties", "method1", 20, 50, 16, 16, 6, 6, 4, 11, 107, "flat"),
]


def main(out=OUT, wide=True, slow=True):
    """slow=False leaves out the cases of SLOW_WIDE_CASES (minutes of reference time each); INDEX.json names them all."""
    os.makedirs(out, exist_ok=True)
    index = []
    for name, method, genes, edges, nc, nt, rows, iters, length, top_k, seed, table in CASES:
        tbl = {"small": small_table(nc, nt, seed), "hyper": None, "flat": np.full((nc + 1, nt + 1), 1.5)}[table]
        p = make_problem(genes, edges, nc, nt, max(rows, 1), length, method=method, top_k=top_k, seed=seed, table=tbl)
        if rows == 0:
            p.perm_cases = np.ones((1, nc + nt), dtype=np.int32)   # unread: iterations == 0
        p.iterations = iters
        dump = os.path.join(out, name + ".txt")
        write_problem(dump, p)
        res = json.loads(subprocess.run([DRIVER, dump, method, str(iters), str(top_k), str(length)], check=True,
                                        capture_output=True, text=True).stdout)
        res["_case"] = {"method": method, "iterations": iters, "top_k": top_k, "path_length": length,
                        "paths": {k: int(v) for k, v in p.levels.n_paths.items()}}
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(res, f, separators=(",", ":"))
        index.append(name)
        print(name, os.path.getsize(dump), "B dump", {k: v for k, v in p.levels.n_paths.items()})
    wide_index = []
    if wide:
        import tempfile
        for name, (method, genes, edges, nc, nt, perms, length, top_k, seed) in WIDE_CASES.items():
            wide_index.append(name)
            if name in SLOW_WIDE_CASES and not slow:
                continue
            p = wide_problem(name)
            with tempfile.TemporaryDirectory() as tmp:
                blob = os.path.join(tmp, name + ".gcrebin")
                write_problem_bin(blob, p)        # tens of MB: the inputs are regenerated from the seed, never committed
                res = json.loads(subprocess.run([DRIVER, "--bin", blob], check=True, capture_output=True, text=True).stdout)
            res["_case"] = {"method": method, "iterations": perms, "top_k": top_k, "path_length": length,
                            "generator": {"genes": genes, "edges": edges, "cases": nc, "ctrls": nt, "seed": seed,
                                          "table": "gcre_values_table",
                                          "table_exact_order": bool(api.values_table_exact_order(nc, nt))},
                            "mask_words": (nc + nt + 63) // 64,
                            "paths": {k: int(v) for k, v in p.levels.n_paths.items()},
                            "input_sha256": problem_digest(p)}
            with open(os.path.join(out, name + ".json"), "w") as f:
                json.dump(res, f, separators=(",", ":"))
            print(name, (nc + nt + 63) // 64, "words", {k: v for k, v in p.levels.n_paths.items()})
    with open(os.path.join(out, "INDEX.json"), "w") as f:
        json.dump({"_provenance": __doc__.strip().splitlines()[0], "cases": index, "wide_cases": wide_index}, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else OUT)
