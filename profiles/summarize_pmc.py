#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counters of one kernel over its dispatches.
usage: profiles/summarize_pmc.py <dir with pass subdirs> <kernel substring> [skip_first_n | -last_n]
A negative third argument keeps only the last n dispatches (the timed launches of a bench run come last: the
calibration launches of the prepass and the warm-up precede them)."""
import csv, glob, json, sys
from collections import defaultdict

def main():
    root, kern = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    out = {}
    for f in sorted(glob.glob(f"{root}/*/*_counter_collection.csv")):
        per = defaultdict(lambda: defaultdict(float))     # dispatch -> counter -> value
        for row in csv.DictReader(open(f)):
            if kern not in row["Kernel_Name"]:
                continue
            per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
        ids = sorted(per)[skip:]          # skip >= 0: drop the first ones; skip < 0: keep the last -skip
        if not ids:
            continue
        names = sorted({c for d in ids for c in per[d]})
        for c in names:
            vals = [per[d][c] for d in ids]
            out[c] = sum(vals) / len(vals)
        out.setdefault("_dispatches", len(ids))
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
