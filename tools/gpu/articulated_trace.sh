# rocprofv3 kernel statistics of the articulated-input configuration (per-part dense product)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/art.py <<'P'
import sys
sys.path.insert(0, '/root/repo')
import torch, bench
dev = torch.device('cuda:0')
r = bench.quick_run(dev, 8, 4096, part_poses=True, steps=6)
print('articulated', r['value'], r['ms_per_step'])
P
rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o p --output-format csv -- python /tmp/art.py 2>&1 | grep articulated
f=$(find /tmp/prof_a -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out && cp $f $GRAFT_REPO_ROOT/gpurun_out/articulated_kernel_stats.csv
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 8.0
for r in rows[:40]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:110]:110s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:7.3f}  per step {float(r['TotalDurationNs'])/1e6/steps:6.2f}")
P
