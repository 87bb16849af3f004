#!/usr/bin/env python3
"""rocprofv3 PMC csv files (FETCH_SIZE, WRITE_SIZE; separate passes) -> profiles/scan_traffic.json.

  python tools/pmc_to_traffic.py FETCH.csv WRITE.csv BENCH.json [--kernel k_scan_rows] [--out FILE]

Mean counter value over the dispatches of the named kernel inside the run; FETCH_SIZE is
doubled as MI355X_MICROARCH.md prescribes for gfx950, both are KiB.  BENCH.json is the
bench line of the same configuration (kernel label, token count, rotation), so that
bench.py can tell whether the file applies to what it is measuring.
"""

import argparse
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mean_counter(path, kernel, counter):
    vals = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.setdefault(row["Dispatch_Id"], 0.0)
                vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    if not vals:
        raise SystemExit("no %s dispatch with %s in %s" % (kernel, counter, path))
    return sum(vals.values()) / len(vals), len(vals)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("bench_json")
    ap.add_argument("--kernel", default="k_scan_rows")
    ap.add_argument("--also", default="k_compact", help="second kernel of the search, reported beside")
    ap.add_argument("--out", default="profiles/scan_traffic.json")
    a = ap.parse_args()
    with open(a.bench_json) as fh:
        bench = json.loads(fh.read().strip().splitlines()[-1])
    fetch, n = mean_counter(a.fetch_csv, a.kernel, "FETCH_SIZE")
    write, _ = mean_counter(a.write_csv, a.kernel, "WRITE_SIZE")
    out = {
        "workload": bench["config"]["workload"].split(":")[0],
        "window": bench["config"]["window"],
        "n_tok": bench["config"]["works_per_gpu"] * bench["config"]["tokens_per_work"],
        "rotate": bench["config"]["distinct_batches"],
        "lanes": bench["config"].get("lanes"),
        "kernel": bench["roofline"]["kernel"],
        "build": __import__("fandom_search_amd._lib", fromlist=["source_hash"]).source_hash(),
        "FETCH_SIZE_KB": fetch,
        "WRITE_SIZE_KB": write,
        "dispatches": n,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                  "(tools/collect_profiles.sh part1), mean over the %s dispatches; FETCH_SIZE doubled as "
                  "MI355X_MICROARCH.md prescribes for gfx950 (the kernel's 64-byte table reads are not wide "
                  "coalesced reads, so the read side is an upper estimate), WRITE_SIZE as reported; KiB" % a.kernel,
    }
    try:
        f2, n2 = mean_counter(a.fetch_csv, a.also, "FETCH_SIZE")
        w2, _ = mean_counter(a.write_csv, a.also, "WRITE_SIZE")
        out["second_kernel"] = {"kernel": a.also, "FETCH_SIZE_KB": f2, "WRITE_SIZE_KB": w2, "dispatches": n2,
                                "hbm_bytes_per_launch": (2.0 * f2 + w2) * 1024.0}
    except SystemExit:
        pass
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
