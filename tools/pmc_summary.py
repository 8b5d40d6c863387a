"""Fold rocprofv3 --pmc passes into profiles/<name>.json (averages per dispatch and kernel).

Usage (on the GPU box, one counter set per pass, nothing but --kernel-trace beside --pmc):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d gpurun_out/pmc/fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d gpurun_out/pmc/write -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -d gpurun_out/pmc/sq1 -- ...
    python3 tools/pmc_summary.py gpurun_out/pmc profiles/r01_pmc_counters.json "note text"
FETCH_SIZE is doubled for the HBM byte figure (gfx950 tallies 64 B per 128-B request for 16 B/lane
reads: MI355X_MICROARCH.md, HBM section); FETCH_SIZE / WRITE_SIZE are in KiB.
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name).strip()


def main(root, out, note):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            cell = acc[k][row["Counter_Name"]]
            cell[0] += float(row["Counter_Value"])
            cell[1] += 1
    kernels = {}
    for k, counters in sorted(acc.items()):
        e = {}
        for c, (total, n) in sorted(counters.items()):
            e[c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = total / max(n, 1)
        e["dispatches"] = max(n for _, n in counters.values())
        if "FETCH_SIZE_KiB" in e and "WRITE_SIZE_KiB" in e:
            e["hbm_bytes"] = (2.0 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024.0
        kernels[k] = e
    json.dump({"note": note, "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in kernels.items():
        print(k, {a: round(b) for a, b in e.items()})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
