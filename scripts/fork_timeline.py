#!/usr/bin/env python3
"""Timeline of the conv-stack launches of one call from a rocprofv3 --kernel-trace CSV: per launch its queue, start and end
relative to the call's first launch, and what the call loses between launches (gaps with nothing running) or wins by overlap.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python scripts/run_once.py f32 1080x1920 8
    python scripts/fork_timeline.py DIR [frames_from_the_end]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
# a call begins with its first conv0 launch; a forked call has two conv0 launches back to back
calls, cur = [], []
for r in rows:
    if "conv0_kernel" in r[2] and cur and not all("conv0_kernel" in k[2] for k in cur):
        calls.append(cur)
        cur = []
    cur.append(r)
if cur:
    calls.append(cur)
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 1
spans = []
for c in calls[1:]:
    t0, t1 = min(k[0] for k in c), max(k[1] for k in c)
    busy, last = 0, t0
    for s, e, *_ in sorted(c):
        if e > last:
            busy += e - max(s, last)
            last = e
    spans.append((t1 - t0, busy, sum(k[1] - k[0] for k in c), len(c)))
for c in calls[-nshow:]:
    t0 = min(k[0] for k in c)
    print(f"call of {len(c)} launches, span {(max(k[1] for k in c) - t0) / 1e3:.1f} us")
    for s, e, name, q, st in c:
        short = name.split("(")[0].replace("void ", "")[:64]
        print(f"   queue {q:>3} stream {st:>3}  {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  {short}")
if spans:
    import statistics
    print(f"{len(spans)} calls: median span {statistics.median(s[0] for s in spans) / 1e3:.1f} us, of which some kernel running "
          f"{statistics.median(s[1] for s in spans) / 1e3:.1f} us (idle between launches {statistics.median(s[0] - s[1] for s in spans) / 1e3:.1f} us); "
          f"sum of the launches' own durations {statistics.median(s[2] for s in spans) / 1e3:.1f} us ({spans[0][3]} launches per call)")
    gaps = [calls[i + 1][0][0] - max(k[1] for k in calls[i]) for i in range(1, len(calls) - 1)]
    if gaps:
        print(f"between calls: median {statistics.median(gaps) / 1e3:.1f} us from the last launch's end to the next call's first start")
