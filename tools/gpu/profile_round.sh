#!/bin/bash
# tools/gpu/profile_round.sh TAG [quick] -- one GPU session on the box (run through gpurun): the GPU test suite, smoke, the
# driver's bench line, rocprofv3 kernel statistics of the same command and the counter passes MI355X_MICROARCH.md
# prescribes (separate --pmc runs, --kernel-trace only).  Everything lands in gpurun_out/TAG/; tools/pmc_traffic.py
# turns the counter CSVs into gpurun_out/TAG/TAG_pmc_traffic.json (copy into profiles/).  `quick` skips the test suite and
# the long bench.
set -u
export TMPDIR=/tmp
TAG=$1
QUICK=${2:-}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
if [ -z "$QUICK" ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest.log 2>&1
  echo "pytest rc $?" >> $O/pytest.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
  timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cp $R/gpurun_out/bench_detail.json $O/bench_detail.json
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2> $O/prof.err
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmc --output-format csv -- python $R/bench.py --plain --steps 1 --warmup 1 > $O/pmc_$tag.log 2>&1
  # the native zpconv ops at the bench workload (the models never call them: their own run, same directory)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmczp --output-format csv -- python $R/tools/zpconv_once.py >> $O/pmc_$tag.log 2>&1
done
# where the wave cycles of the grouping kernels go (verdict r3 item 5): two more passes, SQ counters only
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_breakdown_$i -o pmc --output-format csv -- python $R/bench.py --plain --steps 1 --warmup 1 > $O/pmc_breakdown_$i.log 2>&1
done
cd $R
python tools/pmc_traffic.py $O $O $TAG > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc_breakdown_1 $O/pmc_breakdown_2
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv 2>/dev/null
# the counter CSVs are large: keep the summaries only
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_TCC_HIT_sum_TCC_MISS_sum $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CU_CYCLES_GRBM_GUI_ACTIVE_SQ_WAVES $O/prof
[ -z "$QUICK" ] && tail -5 $O/pytest.log && tail -2 $O/smoke.log
head -12 $O/bench_kernel_stats.csv | cut -c1-160
tail -60 $O/pmc_summary.txt
