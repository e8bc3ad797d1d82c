#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace: per-stream busy time, idle gaps on the busiest stream and the kernels that follow the
largest gaps -- for the last TIMED step (as the bench runs it: composite calls, side stream) and for the last step of the process
(bench.py's instrumented roofline step: every GEMM launched on its own between two HIP events, side stream off)."""
import csv
import glob
import sys


def main():
    d, nsteps = sys.argv[1], int(sys.argv[2])
    rows = list(csv.DictReader(open(glob.glob(d + "/*kernel_trace.csv")[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # split steps by adamw launches
    ends = [i for i, r in enumerate(rows) if "sumsq_final" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
    if len(ends) >= 4:
        analyse(rows[ends[-4] + 1:ends[-3] + 1], "last timed step")
    analyse(rows[ends[-2] + 1:ends[-1] + 1], "last step (instrumented)")


def analyse(step, label):
    t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
    print(f"{label}: {len(step)} kernels, span {(t1 - t0) / 1e6:.3f} ms")
    streams = {}
    for r in step:
        streams.setdefault((r["Queue_Id"], r.get("Stream_Id", "")), []).append(r)
    for k, v in sorted(streams.items(), key=lambda kv: -len(kv[1])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
        print(f"  queue/stream {k}: {len(v)} kernels, busy {busy / 1e6:.3f} ms")
    main_q = max(streams.items(), key=lambda kv: len(kv[1]))[1]
    gaps = []
    prev_end = int(main_q[0]["Start_Timestamp"])
    for r in main_q:
        g = int(r["Start_Timestamp"]) - prev_end
        if g > 0:
            gaps.append((g, r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e6))
        prev_end = max(prev_end, int(r["End_Timestamp"]))
    tot = sum(g for g, _, _ in gaps)
    print(f"  main queue idle gaps: total {tot / 1e6:.3f} ms in {len(gaps)} gaps; >5us: {sum(1 for g, _, _ in gaps if g > 5000)}")
    for g, n, t in sorted(gaps, reverse=True)[:12]:
        print(f"    gap {g / 1e3:8.1f} us before {n} at t={t:.2f} ms")
    # union busy over all queues
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print(f"  GPU busy (union of all queues): {u / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms")
    # the tail of backward: what each queue runs in the last 1.2 ms before the gradient norm (which waits for all of them)
    sq = next((r for r in step if "sumsq_kernel" in r["Kernel_Name"]), None)
    if sq is not None:
        ts = int(sq["Start_Timestamp"])
        print("  the last 1.2 ms before the gradient norm (queue: kernel, start .. end in us before it):")
        for r in step:
            e = int(r["End_Timestamp"])
            if e <= ts and ts - e < 1200000:
                print(f"    q{r['Queue_Id']}  {(ts - int(r['Start_Timestamp'])) / 1e3:7.1f} .. {(ts - e) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    if len(sys.argv) > 3:      # every dispatch of one kernel in the step: duration, grid, what ran before it
        key = sys.argv[3]
        for i, r in enumerate(step):
            if key in r["Kernel_Name"]:
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                prev = step[i - 1]["Kernel_Name"][:50] if i else "-"
                print(f"    t={(int(r['Start_Timestamp']) - t0) / 1e6:7.3f} ms  {dur:7.1f} us  grid {r.get('Grid_Size_X', '?'):>9}  {r['Kernel_Name'][:110]}  <- {prev}")


if __name__ == "__main__":
    main()
