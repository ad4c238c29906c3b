#!/usr/bin/env python3
"""Generate tests/golden/* from the reference checkout (run in the build container only).

The fixtures are DATA: the reference tests' input files and the expected
tables those tests compare against.  No reference source text is stored.

Sources (all under /root/reference):
  tests/data/{overlap,nearest,count_overlaps}/{reads,targets}.csv  -> inputs, copied
  tests/_expected.py:10-128   PD_DF_OVERLAP         -> expected_overlap.csv
  tests/_expected.py:130-172  PD_DF_NEAREST         -> expected_nearest.csv
  tests/_expected.py:183-202  PD_DF_COUNT_OVERLAPS  -> expected_count_overlaps.csv
  tests/_expected.py:174-181  PD_DF_MERGE           -> expected_merge.csv
  tests/data/merge/input.csv, tests/data/coverage/{reads,targets}.csv -> inputs, copied
  tests/test_partitioned_range_operation_regressions.py:24-59,128-190 -> cases.json (sort_scan)
  tests/test_coordinate_system_metadata.py:1032-1055,1577-1623 -> cases.json (sort_scan_boundary)
  tests/data/exons/*.parquet, tests/data/fBrain-DS14718/*.parquet -> copied
      (known answer 54,246 overlaps 0-based: docs/supplement.md:108,111,149)
  tests/test_coordinate_system_metadata.py:738-819,1172-1191,1482-1506 -> cases.json (boundary)
  tests/test_overlap_output_mode.py:20-46,99-119    -> cases.json (output modes)
  docs/notebooks/tutorial.ipynb cells 4,9,13,17     -> cases.json (tutorial)

The expected tables are extracted by parsing the dict literals in
tests/_expected.py with ``ast`` (the module itself needs polars, absent here).
"""
import ast
import csv
import json
import os
import shutil
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract_tables():
    src = open(os.path.join(REF, "tests/_expected.py")).read()
    tree = ast.parse(src)
    want = {"PD_DF_OVERLAP": "expected_overlap.csv",
            "PD_DF_NEAREST": "expected_nearest.csv",
            "PD_DF_COUNT_OVERLAPS": "expected_count_overlaps.csv",
            "PD_DF_MERGE": "expected_merge.csv"}
    done = set()
    for node in tree.body:
        if not isinstance(node, ast.Assign) or len(node.targets) != 1:
            continue
        name = getattr(node.targets[0], "id", None)
        if name not in want or name in done:
            continue
        # first assignment is pd.DataFrame({...}).astype({...}); dig out the dict literal
        dicts = [n for n in ast.walk(node.value) if isinstance(n, ast.Dict)]
        table = None
        for d in dicts:
            try:
                val = ast.literal_eval(d)
            except Exception:
                continue
            if val and all(isinstance(v, list) for v in val.values()):
                table = val
                break
        if table is None:
            continue
        done.add(name)
        cols = list(table)
        rows = list(zip(*[table[c] for c in cols]))
        with open(os.path.join(OUT, want[name]), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols)
            w.writerows(rows)
        print(f"{want[name]}: {len(rows)} rows")
    assert done == set(want), done


def copy_inputs():
    for op in ("overlap", "nearest", "count_overlaps"):
        os.makedirs(os.path.join(OUT, op), exist_ok=True)
        for f in ("reads.csv", "targets.csv"):
            shutil.copyfile(os.path.join(REF, "tests/data", op, f), os.path.join(OUT, op, f))
    os.makedirs(os.path.join(OUT, "merge"), exist_ok=True)
    shutil.copyfile(os.path.join(REF, "tests/data/merge/input.csv"), os.path.join(OUT, "merge/input.csv"))
    os.makedirs(os.path.join(OUT, "coverage"), exist_ok=True)
    for f in ("reads.csv", "targets.csv"):
        shutil.copyfile(os.path.join(REF, "tests/data/coverage", f), os.path.join(OUT, "coverage", f))
    for d in ("exons", "fBrain-DS14718"):
        os.makedirs(os.path.join(OUT, d), exist_ok=True)
        for f in os.listdir(os.path.join(REF, "tests/data", d)):
            if f.endswith(".parquet"):
                shutil.copyfile(os.path.join(REF, "tests/data", d, f), os.path.join(OUT, d, f))


def write_cases():
    iv = lambda rows: {"chrom": [r[0] for r in rows], "start": [r[1] for r in rows], "end": [r[2] for r in rows]}
    one = lambda s, e: iv([("chr1", s, e)])
    cases = {
        "_source": "restated from the reference tests named in each case; values only",
        "boundary_overlap": [
            # tests/test_coordinate_system_metadata.py:738-819
            {"name": "adjacent_zero_based", "zero_based": True, "df1": one(100, 200), "df2": one(200, 300), "n_pairs": 0},
            {"name": "adjacent_one_based", "zero_based": False, "df1": one(100, 200), "df2": one(200, 300), "n_pairs": 1},
            {"name": "touching_zero_based", "zero_based": True, "df1": one(100, 200), "df2": one(199, 300), "n_pairs": 1},
            {"name": "gap_one_based", "zero_based": False, "df1": one(100, 200), "df2": one(202, 300), "n_pairs": 0},
            {"name": "same_zero_based", "zero_based": True, "df1": one(100, 200), "df2": one(100, 200), "n_pairs": 1},
            {"name": "same_one_based", "zero_based": False, "df1": one(100, 200), "df2": one(100, 200), "n_pairs": 1},
            {"name": "contained_zero_based", "zero_based": True, "df1": one(100, 200), "df2": one(150, 180), "n_pairs": 1},
            {"name": "contained_one_based", "zero_based": False, "df1": one(100, 200), "df2": one(150, 180), "n_pairs": 1},
        ],
        "boundary_count": [
            # tests/test_coordinate_system_metadata.py:1172-1191
            {"name": "count_adjacent_zero_based", "zero_based": True, "df1": one(100, 200), "df2": one(200, 300), "counts": [0]},
            {"name": "count_adjacent_one_based", "zero_based": False, "df1": one(100, 200), "df2": one(200, 300), "counts": [1]},
            # tests/test_coordinate_system_metadata.py:1482-1506 (UInt32 columns, df1 order kept)
            {"name": "count_uint32", "zero_based": True, "dtype": "uint32",
             "df1": iv([("chr1", 100, 150), ("chr1", 200, 250), ("chr1", 300, 350)]),
             "df2": iv([("chr1", 125, 175), ("chr1", 225, 275)]), "counts": [1, 1, 0]},
        ],
        "output_mode": {
            # tests/test_overlap_output_mode.py:20-46 (inputs), :99-119 (expected), 0-based
            "zero_based": True,
            "df1": {"chrom": ["chr1", "chr1", "chr1", "chr2"], "start": [100, 100, 1000, 50],
                    "end": [200, 200, 1100, 60], "name": ["dup", "dup", "miss", "other"]},
            "df2": {"chrom": ["chr1", "chr1", "chr2"], "start": [90, 120, 55], "end": [150, 180, 56],
                    "score": [1, 2, 3]},
            "left": {"chrom": ["chr1", "chr1", "chr1", "chr1", "chr2"], "start": [100, 100, 100, 100, 50],
                     "end": [200, 200, 200, 200, 60], "name": ["dup", "dup", "dup", "dup", "other"]},
            "left_distinct": {"chrom": ["chr1", "chr1", "chr2"], "start": [100, 100, 50],
                              "end": [200, 200, 60], "name": ["dup", "dup", "other"]},
        },
        "tutorial": {
            # docs/notebooks/tutorial.ipynb cells 4, 9, 13, 17 (1-based / Weak)
            "zero_based": False,
            "df1": iv([("chr1", 1, 5), ("chr1", 3, 8), ("chr1", 8, 10), ("chr1", 12, 14)]),
            "df2": iv([("chr1", 4, 8), ("chr1", 10, 11)]),
            "overlap": [[1, 5, 4, 8], [3, 8, 4, 8], [8, 10, 4, 8], [8, 10, 10, 11]],
            "nearest": [[1, 5, 4, 8, 0], [3, 8, 4, 8, 0], [8, 10, 4, 8, 0], [12, 14, 10, 11, 1]],
            "count": [1, 1, 2, 0],
        },
        "sort_scan": {
            # tests/test_partitioned_range_operation_regressions.py: inputs :128-160, view :178-186,
            # expected tables :24-59 (0-based)
            "zero_based": True,
            "left": iv([("chr1", 0, 10), ("chr1", 20, 30), ("chr1", 8, 25)]),
            "right": iv([("chr1", 5, 10), ("chr1", 20, 25)]),
            "view": iv([("chr1", 0, 40)]),
            "merge": {"start": [0], "end": [30], "n_intervals": [3]},
            "complement": {"start": [30], "end": [40]},
            "subtract": {"start": [0, 10, 25], "end": [5, 20, 30]},
            "cluster": {"start": [0, 8, 20], "end": [10, 25, 30], "cluster": [0, 0, 0], "cluster_start": [0, 0, 0],
                        "cluster_end": [30, 30, 30]},
        },
        "sort_scan_boundary": [
            # tests/test_coordinate_system_metadata.py:1032-1055 (merge: adjacent intervals) and :1577-1623
            # (coverage: UInt32 boundary) -- the reference's own pins of the Weak / Strict rule for these operations
            {"name": "merge_adjacent_zero_based", "op": "merge", "zero_based": True,
             "df": iv([("chr1", 100, 150), ("chr1", 150, 200)]), "n_rows": 2},
            {"name": "merge_adjacent_one_based", "op": "merge", "zero_based": False,
             "df": iv([("chr1", 100, 150), ("chr1", 150, 200)]), "n_rows": 1},
            {"name": "coverage_boundary_zero_based", "op": "coverage", "zero_based": True, "dtype": "uint32",
             "df1": one(100, 200), "df2": one(200, 300), "coverage": [0]},
            {"name": "coverage_boundary_one_based", "op": "coverage", "zero_based": False, "dtype": "uint32",
             "df1": one(100, 200), "df2": one(200, 300), "coverage": [1]},
        ],
        "known_answers": {
            # docs/supplement.md:108,111,149 -- exons (df1) x fBrain (df2), 0-based
            "exons_x_fbrain_strict_pairs": 54246,
        },
    }
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference checkout not mounted; fixtures are already committed")
    copy_inputs()
    extract_tables()
    write_cases()
