"""Aggregate a rocprofv3 PC-sampling run (csv) into a small histogram: samples per (code object, offset, instruction,
stall reason ...) -- every column that is not an id / timestamp / per-wave location.  The raw csv of a 1 s kernel is
hundreds of MB; this runs on the GPU box and only the histogram (and the first rows, to document the format) travel.
usage: python scripts/pc_sample_hist.py <rocprofv3 output dir> <out prefix>"""
import collections
import csv
import glob
import json
import os
import sys

DROP = ("timestamp", "dispatch", "correlation", "exec_mask", "wave", "chiplet", "hw_id", "workgroup", "cu_id", "simd",
        "slot", "sample", "thread", "queue", "agent", "pid", "tid", "xcc", "vmid", "se_id", "pipe")


def main():
    root, out = sys.argv[1], sys.argv[2]
    files = [f for f in glob.glob(os.path.join(root, "**", "*"), recursive=True) if os.path.isfile(f)]
    with open(out + "_files.txt", "w") as fh:
        for f in files:
            fh.write("%s %d\n" % (f, os.path.getsize(f)))
    for f in files:
        if not f.endswith(".csv") or "pc_sampl" not in os.path.basename(f):
            continue
        tag = os.path.basename(f).replace(".csv", "")
        with open(f, newline="") as fh:
            head = [next(fh, "") for _ in range(40)]
        open(out + "_" + tag + "_head.txt", "w").write("".join(head))
        hist = collections.Counter()
        n = 0
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            keep = [c for c in (rd.fieldnames or []) if not any(d in c.lower() for d in DROP)]
            for row in rd:
                hist[tuple(row[c] for c in keep)] += 1
                n += 1
        rows = [{"n": v, **dict(zip(keep, k))} for k, v in hist.most_common(6000)]
        json.dump({"file": os.path.basename(f), "samples": n, "columns": keep, "distinct": len(hist), "rows": rows},
                  open(out + "_" + tag + "_hist.json", "w"))
        print(tag, "samples", n, "distinct", len(hist), "columns", keep)


if __name__ == "__main__":
    main()
