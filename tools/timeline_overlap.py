"""Measurement tooling: from a rocprofv3 --kernel-trace CSV, print for the last training step when each AdamW range launch
started / ended relative to the step's first kernel, and which kernels ran beside it (stream overlap check).

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline ...
  python tools/timeline_overlap.py gpurun_out/tl"""
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    ad = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    if not ad:
        print("no adamw kernels")
        return
    # the last step's AdamW launches: the trailing run of adamw kernels separated by < 20 ms
    last = [ad[-1]]
    for i in reversed(ad[:-1]):
        if rows[last[0]][0] - rows[i][0] < 8_000_000:
            last.insert(0, i)
        else:
            break
    t_end = max(rows[i][1] for i in last)
    prev_ad_end = max((rows[i][1] for i in ad if rows[i][1] < rows[last[0]][0] - 8_000_000), default=rows[0][0])
    print(f"step window: {(t_end - prev_ad_end) / 1e6:.3f} ms (end of previous step's last AdamW -> end of this step's last)")
    for i in last:
        s, e, n, q = rows[i]
        beside = [(r[2][:60], r[3]) for r in rows if r[0] < e and r[1] > s and r is not rows[i]]
        print(f"adamw q{q}: start {(s - prev_ad_end) / 1e6:7.3f} ms, {(e - s) / 1e3:7.1f} us, beside {len(beside)} kernels: "
              + ", ".join(sorted({b[0].split('(')[0][-40:] for b in beside}))[:200])
    qs = {}
    for r in rows:
        if prev_ad_end <= r[0] <= t_end:
            qs.setdefault(r[3], [0, 0.0])
            qs[r[3]][0] += 1
            qs[r[3]][1] += (r[1] - r[0]) / 1e6
    for q, (n, ms) in sorted(qs.items()):
        print(f"queue {q}: {n} kernels, {ms:.3f} ms busy")


if __name__ == "__main__":
    main()
