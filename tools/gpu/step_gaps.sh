# idle time of the GPU between kernels in the steady-state steps of `bench.py --plain`
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_g -o p --output-format csv -- python /root/repo/bench.py --plain --steps 6 --warmup 2 > /tmp/g.log 2>&1
tail -c 400 /tmp/g.log
t=$(find /tmp/prof_g -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/step_gaps.py $t
