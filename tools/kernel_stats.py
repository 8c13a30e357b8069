"""rocprofv3 --kernel-trace output directory -> per-kernel duration table (CSV on stdout).
With MF_MARK=<kernel substring> only the launches between the 2nd and 3rd occurrence of that kernel count
(bench.py's MF_BENCH_MARK brackets its timed steps with k_icc_scene_setup launches)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0][:100]


def load(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def main():
    rows = load(sys.argv[1])
    mark = os.environ.get("MF_MARK")
    note = "all launches"
    if mark:
        marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
        if len(marks) >= 3:
            rows = rows[marks[1] + 1: marks[2]]
            note = f"launches between the 2nd and 3rd {mark}"
    if os.environ.get("MF_SEQ"):  # the launches in order: index, start offset (us), duration (us), name
        t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
        with open(os.environ["MF_SEQ"], "w") as f:
            for i, r in enumerate(rows):
                f.write(f'{i},{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},'
                        f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f},{short(r["Kernel_Name"])}\n')
    agg = collections.OrderedDict()
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print(f"# rocprofv3 --kernel-trace; {note}; sum of kernel durations {total / 1e6:.3f} ms")
    print("Kernel,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percent")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{a[2]},{a[3]},{100 * a[1] / total:.2f}')


if __name__ == "__main__":
    main()
