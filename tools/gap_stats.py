"""GPU idle time between the kernels of a rocprofv3 --kernel-trace (csv): how much of a step is launch gaps rather than kernels.
Usage: python tools/gap_stats.py <kernel_trace.csv> <first-kernel-of-a-step regex> [skip_steps]
Prints per step (median over the steps after `skip_steps`): wall between step starts, sum of kernel durations, idle, launches, and the
largest gaps with the kernels on either side."""
import csv, re, sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pat = re.compile(sys.argv[2])
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    starts = [i for i, e in enumerate(ev) if pat.search(e[2])]
    steps = []
    gaps = defaultdict(list)
    for a, b in zip(starts[skip:-1], starts[skip + 1:]):
        seg = ev[a:b + 1]
        wall = seg[-1][0] - seg[0][0]
        busy = sum(e[1] - e[0] for e in seg[:-1])
        idle = 0
        for p, q in zip(seg[:-1], seg[1:]):
            g = max(0, q[0] - p[1])
            idle += g
            gaps[(p[2][:50], q[2][:50])].append(g)
        steps.append((wall, busy, idle, len(seg) - 1))
    if not steps:
        print("no steps found")
        return
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"steps {len(steps)}: wall {med([s[0] for s in steps]) / 1e3:.1f} us, kernels {med([s[1] for s in steps]) / 1e3:.1f} us, "
          f"idle {med([s[2] for s in steps]) / 1e3:.1f} us, launches {med([s[3] for s in steps])}")
    top = sorted(((sum(v) / len(steps), k) for k, v in gaps.items()), reverse=True)[:14]
    for g, (p, q) in top:
        print(f"  {g / 1e3:7.1f} us/step idle between  {p}  ->  {q}")


main()
