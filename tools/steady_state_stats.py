#!/usr/bin/env python3
"""Per-kernel time of the LAST `--window_ms` milliseconds of a `rocprofv3 --kernel-trace --output-format csv` trace.

A benchmark's warm-up (MIOpen's find mode times every convolution algorithm, start-up self-checks, graph captures)
dominates `--stats` output; the timed steps are the tail of the trace.  Usage:

    python tools/steady_state_stats.py <..._kernel_trace.csv> --window_ms 300 --steps 10 [--top 40]

prints one line per kernel name: ms per step, calls per step, share of the window's kernel time."""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--window_ms", type=float, required=True)
ap.add_argument("--steps", type=int, required=True, help="steps inside the window (for the per-step columns)")
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()

rows = []
with open(a.trace, newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
end = max(r[1] for r in rows)
lo = end - int(a.window_ms * 1e6)
tot, calls = collections.Counter(), collections.Counter()
for s, e, n in rows:
    if s >= lo:
        tot[n] += e - s
        calls[n] += 1
busy = sum(tot.values())
print(f"window {a.window_ms:.1f} ms, kernel time {busy / 1e6:.1f} ms ({100 * busy / (a.window_ms * 1e6):.0f} % of the window), "
      f"{sum(calls.values()) / a.steps:.0f} launches per step")
print(f"{'ms/step':>9} {'calls/step':>10} {'share':>6}  kernel")
for n, t in tot.most_common(a.top):
    print(f"{t / 1e6 / a.steps:9.3f} {calls[n] / a.steps:10.1f} {100 * t / busy:5.1f}%  {n[:150]}")
