#!/bin/bash
# Round profiles on the GPU box (run from the repo root through gpurun):
#   bash tools/collect_profiles.sh r02
# kernel-trace statistics of the bench command (fp64 headline, FLOAT32, INT8) and the PMC passes the MI355X guide
# prescribes (one counter group per run; FETCH_SIZE and WRITE_SIZE cannot share a pass), folded by
# tools/pmc_summary.py.  Outputs land in gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in float64 float32 int8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -o t -- python3 $ROOT/bench.py --mode $mode --steps 250 --warmup 50 --no-cpu-baseline > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err || exit 1
done
pmc() {   # mode, name, counters...
  local mode=$1 name=$2; shift 2
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_${mode}_$name -o p -- python3 $ROOT/bench.py --mode $mode --steps 8 --warmup 2 --no-cpu-baseline > $OUT/pmc_${mode}_$name.log 2>&1 || exit 1
  echo "pmc $mode $name done"
}
pmc float64 FETCH FETCH_SIZE
pmc float64 WRITE WRITE_SIZE
pmc float64 SQ SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc float64 GRBM GRBM_GUI_ACTIVE
pmc int8 LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pmc int8 FETCH FETCH_SIZE
pmc int8 WRITE WRITE_SIZE
pmc float32 LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_float64.json $OUT/pmc_float64_FETCH $OUT/pmc_float64_WRITE $OUT/pmc_float64_SQ $OUT/pmc_float64_GRBM
python3 tools/pmc_summary.py $OUT/pmc_int8.json $OUT/pmc_int8_LDS $OUT/pmc_int8_FETCH $OUT/pmc_int8_WRITE
python3 tools/pmc_summary.py $OUT/pmc_float32.json $OUT/pmc_float32_LDS
ls $OUT
