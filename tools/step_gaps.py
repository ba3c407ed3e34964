"""tools/step_gaps.py KERNEL_TRACE.csv [N_LAST_KERNELS_FRACTION] -- idle time of the GPU between consecutive kernels of a
`rocprofv3 --kernel-trace` run of `bench.py --plain`: total busy / idle time over the last third of the trace (steady-state
steps), the largest gaps with the kernels on either side, and idle time summed by the kernel that FOLLOWS the gap (who was
late)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda x: x[0])
ks = ks[len(ks) * 2 // 3:]
busy = sum(e - s for s, e, _ in ks)
span = ks[-1][1] - ks[0][0]
gaps = []
cur_end = ks[0][1]
for i in range(1, len(ks)):
    s, e, n = ks[i]
    if s > cur_end:
        gaps.append((s - cur_end, ks[i - 1][2], n))
    cur_end = max(cur_end, e)
idle = sum(g for g, _, _ in gaps)
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '')[:70]
print(f'kernels {len(ks)}  span {span / 1e6:.2f} ms  sum of kernel durations {busy / 1e6:.2f} ms  idle between kernels {idle / 1e6:.2f} ms ({100.0 * idle / span:.1f} %)')
print('largest gaps (us): before <- after')
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f'  {g / 1e3:8.1f}  {short(a)}  ->  {short(b)}')
by = collections.Counter()
cnt = collections.Counter()
for g, a, b in gaps:
    by[short(b)] += g
    cnt[short(b)] += 1
print('idle time by the kernel that follows (us total, count):')
for n, g in by.most_common(20):
    print(f'  {g / 1e3:8.1f}  {cnt[n]:4d}  {n}')
