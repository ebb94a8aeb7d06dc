#!/bin/bash
# HBM traffic and duration of the tensor-level hook kernels on an 8192 x 8192 fp32 tensor (run through gpurun from the
# repo root): one kernel-trace pass and one --pmc pass per counter, as the MI355X guide prescribes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hookprof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/tools/hook_timing.py 8192 > $OUT/trace.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python3 $ROOT/tools/hook_timing.py 8192 > $OUT/pmc_$c.log 2>&1 || exit 1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat > $OUT/kernel_stats.csv
