"""Summarise a per-wave start / end trace of k_horizon (probe build -DHZ_WG_TRACE, env HZ_WG_TRACE_OUT; hz_horizon.hip):
wave lifetimes, how many waves are resident over time, and how long the launch's tail is -- the time after the LAST wave
has started (nothing left to hand out) until the launch ends.
usage: python scripts/wg_trace_stats.py <trace.txt[.gz]> [out.json]"""
import gzip
import json
import sys

import numpy as np


def launches(path):
    op = gzip.open if path.endswith(".gz") else open
    cur, hdr = [], None
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                if cur:
                    yield hdr, np.array(cur, np.int64)
                hdr, cur = line[1:].strip(), []
            else:
                a = line.split()
                if len(a) == 3:
                    cur.append((int(a[0]), int(a[1]), int(a[2])))
    if cur:
        yield hdr, np.array(cur, np.int64)


def stats(hdr, t):
    tick = 1e-5                                   # 100 MHz counter -> ms
    s, e, x = t[:, 0] * tick, t[:, 1] * tick, t[:, 2]
    life = e - s
    end = e.max()
    last_start = s.max()
    # resident waves over time (sampled)
    grid = np.linspace(0.0, end, 201)
    starts, ends = np.sort(s), np.sort(e)
    resident = np.searchsorted(starts, grid, side="right") - np.searchsorted(ends, grid, side="right")
    full = resident.max()
    # ideal: the same wave-milliseconds at full residency
    area = life.sum()
    ideal = area / full
    per_xcd = {int(k): {"waves": int((x == k).sum()), "end_ms": float(e[x == k].max()), "last_start_ms": float(s[x == k].max())}
               for k in np.unique(x)}
    return {"launch": hdr, "waves": int(len(s)), "end_ms": float(end), "last_wave_started_ms": float(last_start),
            "tail_ms_after_last_start": float(end - last_start),
            "wave_lifetime_ms": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "median": float(np.median(life)),
                                 "p90": float(np.percentile(life, 90)), "max": float(life.max())},
            "max_resident_waves": int(full), "wave_ms_total": float(area), "ideal_ms_at_full_residency": float(ideal),
            "efficiency_vs_full_residency": float(ideal / end),
            "resident_waves_at_fraction_of_launch": {"%.2f" % f: int(resident[int(round(f * 200))]) for f in (0.1, 0.5, 0.8, 0.9, 0.95, 0.98)},
            "per_xcd": per_xcd}


def main():
    out = [stats(h, t) for h, t in launches(sys.argv[1])]
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
