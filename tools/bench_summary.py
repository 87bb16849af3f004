#!/usr/bin/env python3
"""One line per artifact under profiles/ (or the files named): the figures DESIGN.md section 8
quotes.  `python tools/bench_summary.py [profiles/r03_*.json ...]`"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_line(f, d):
    r = d["roofline"]
    out = [os.path.basename(f), "value %.4g" % d["value"], "us/step %.2f" % (d["ms_per_step"] * 1e3),
           "step %.3f" % r.get("step", 0), "frac %.3f" % r["frac"],
           "alone_us %.2f" % (r.get("launch_ms_alone", 0) * 1e3), "frac_alone %.3f" % r.get("frac_alone", 0),
           r["bound"], r["kernel"], "traffic %s" % (r["traffic"] and round(r["traffic"] / 1e6, 1))]
    if "samples_ms" in d:
        out.append("samples_us/step %s" % [round(x * 1e3 / d["steps"], 1) for x in d["samples_ms"]])
    print("  ".join(str(x) for x in out))
    for k, v in d.get("companions", {}).items():
        print("    companion %-22s ms_per_step %s  value %s" % (k, v.get("ms_per_step"), v.get("value")))
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        print("    cpu_baseline %.1f %s on %d cores; reference-shaped %.1f" %
              (c["value"], c["unit"], c["cores"], c.get("reference_shaped", {}).get("value", 0)))


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*")) +
                                   [os.path.join(ROOT, "profiles", "scan_traffic.json")])
    for f in files:
        if f.endswith(".build.json") or f.endswith(".txt"):
            continue
        try:
            text = open(f).read().strip()
        except OSError as e:
            print(f, "ERR", e)
            continue
        if f.endswith("kernel_stats.csv"):
            rows = list(csv.DictReader(text.splitlines()))[:6]
            print(os.path.basename(f) + "  " + "  ".join(
                "%s %.1f us x%s" % ((re.search(r"(k_\w+)", r["Name"]) or re.search(r"(\w+)", r["Name"])).group(1),
                                    float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows))
            continue
        if f.endswith(".csv"):
            continue
        try:
            try:
                d = json.loads(text)
            except ValueError:
                d = json.loads(text.splitlines()[-1])
        except ValueError:
            print(os.path.basename(f) + "  " + " | ".join(l[:160] for l in text.splitlines() if l.startswith("{") or "ms" in l)[:900])
            continue
        if isinstance(d, dict) and "roofline" in d and "value" in d:
            bench_line(f, d)
        elif os.path.basename(f).startswith("scan_traffic") or "hbm_bytes_per_launch" in d:
            d = json.loads(text)
            print("%s  %s lanes %s: %.1f MB per launch (algorithmic %.1f MB), second kernel %s MB  build %s" % (
                os.path.basename(f), d["kernel"], d["lanes"], d["hbm_bytes_per_launch"] / 1e6,
                d["algorithmic_bytes_per_launch"] / 1e6,
                round(d.get("second_kernel", {}).get("hbm_bytes_per_launch", 0) / 1e6, 1), d.get("build")))
        else:
            print(os.path.basename(f) + "  " + json.dumps(d)[:700])


if __name__ == "__main__":
    main()
