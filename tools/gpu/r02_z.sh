#!/bin/bash
# round 2, GPU session Z (final build): full GPU suite on the final build, bench with every configuration, rocprofv3 stats,
# HBM traffic counters of the bench step.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
( time timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 ) > gpurun_out/r02z/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02z/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02z/smoke.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r02z/bench.json 2> gpurun_out/r02z/bench.err
timeout 600 python bench.py --steps 5 --warmup 2 --separable --no-cpu-baseline > gpurun_out/r02z/bench_separable_fwd.json 2> gpurun_out/r02z/bench_separable_fwd.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02z/prof -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $R/gpurun_out/r02z/bench_under_rocprof.json 2> $R/gpurun_out/r02z/prof.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/r02z/pmc_$tag -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/r02z/pmc_$tag.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $R/gpurun_out/r02z/pmc_mfma -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/r02z/pmc_mfma.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/r02z/pmc_waves -o pmc --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs > $R/gpurun_out/r02z/pmc_waves.log 2>&1
cd $R
tail -9 gpurun_out/r02z/pytest.log; cat gpurun_out/r02z/smoke.log | tail -2; ls gpurun_out/r02z/prof | head
