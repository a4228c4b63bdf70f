#!/usr/bin/env python3
"""joins bench.py's `kernel_symbols` (HIP-event medians of the roofline leg) with a rocprofv3 kernel_stats.csv of
`rocprofv3 --kernel-trace --stats --mangled-kernels -- python3 bench.py --serial-streams ...` (one stream: isolated launches) through
the mangled-name pattern every symbol row carries, and prints the two average durations side by side.

    python scripts/compare_bench_rocprof.py bench_line.json kernel_stats.csv
"""
import csv
import fnmatch
import json
import sys


def main():
    with open(sys.argv[1]) as f:
        line = [l for l in f if l.startswith('{"metric"')][-1]
    out = json.loads(line)
    stats = list(csv.DictReader(open(sys.argv[2])))
    print(f"# bench: {out['value']} images/s, {out['ms_per_step']} ms/step; roofline kernel {out['roofline']['kernel']} "
          f"{out['roofline']['avg_launch_us']} us x {out['roofline']['launches_per_step']} = frac {out['roofline']['frac']}"
          f"{' SUSPECT' if out['roofline'].get('suspect') else ''}")
    print(f"{'symbol':44s} {'n/step':>6s} {'bench avg us':>12s} {'min':>8s} {'median':>8s} {'max':>8s} {'rocprof avg us':>14s} {'calls':>6s} {'ratio':>6s}")
    for sym, r in sorted(out["kernel_symbols"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
        pat = "*" + r["rocprof_mangled"] + "*"
        rows = [s for s in stats if fnmatch.fnmatchcase(s["Name"], pat)]
        calls = sum(int(s["Calls"]) for s in rows)
        avg = sum(float(s["TotalDurationNs"]) for s in rows) / calls / 1e3 if calls else float("nan")
        print(f"{sym:44s} {r['launches_per_step']:6d} {r['avg_launch_us']:12.2f} {r['min_us']:8.2f} {r['median_us']:8.2f} {r['max_us']:8.2f} "
              f"{avg:14.2f} {calls:6d} {r['avg_launch_us'] / avg if calls else float('nan'):6.3f}")


if __name__ == "__main__":
    main()
