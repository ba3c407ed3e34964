# kernel statistics of the native zpconv backward alone (8 x 4096, C = 64)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bwd_only.py <<'P'
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/equi-articulated-pose_amd')
import torch, synth_clouds
import vgtk.cuda.zpconv as Z
import vgtk.cuda.grouping as G
B, P, A, K, NN, C = 8, 4096, 60, 24, 64, 64
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
ball = G.ball_query(xyz, xyz, synth_clouds.backbone_layers(P)[1][2], NN)
idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
w = torch.rand(B, P, A, K, NN, device=dev)
g = torch.randn(B, C, K, P, A, device=dev)
for _ in range(2):
    Z.inter_zpconv_backward(idx, w, g, P)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    Z.inter_zpconv_backward(idx, w, g, P)
e1.record(); torch.cuda.synchronize()
print('backward ms', e0.elapsed_time(e1) / 5)
P
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o p --output-format csv -- python /tmp/bwd_only.py 2>&1 | grep "backward ms"
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms  tot {float(r['TotalDurationNs'])/1e6:8.2f}")
P
t=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
python - "$t" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call's timeline
names = [r['Kernel_Name'] for r in rows]
last = max(i for i, n in enumerate(names) if 'first_rows' in n)
t0 = int(rows[last]['Start_Timestamp'])
for r in rows[last:last + 14]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} .. {(int(r['End_Timestamp']) - t0) / 1e6:8.3f} ms  {r['Kernel_Name'][:60]}")
P
