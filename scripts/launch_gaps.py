"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (the cost a hipGraph could remove).
Usage: launch_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
rows = [r for r in rows if "synth" not in r[2] and "probe" not in r[2] and "copyBuffer" not in r[2]]
busy = sum(e - s for s, e, _ in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
inner = [g for g in gaps if 0 <= g < 2_000_000]          # gaps inside the timed loop (not between phases of the script)
print(f"{len(rows)} kernels, busy {busy / 1e6:.2f} ms, {len(inner)} gaps between consecutive kernels: "
      f"sum {sum(inner) / 1e6:.3f} ms = {100.0 * sum(inner) / busy:.3f} % of the busy time, "
      f"median {sorted(inner)[len(inner) // 2] / 1e3:.1f} us, max {max(inner) / 1e3:.1f} us")
