#!/bin/bash
# Round evidence on the GPU box: bench line, the same command under rocprofv3 (kernel stats), the two PMC
# traffic passes and the SQ counter passes.  usage: tools/evidence.sh <round tag, e.g. r02> <commit>
# Writes summaries under gpurun_out/<tag>_evidence/ ; copy what is to be judged into profiles/.
set -o pipefail
TAG=$1; COMMIT=$2; OUT=gpurun_out/${TAG}_evidence; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# PMC traffic first: bench.py cites profiles/<tag>_pmc_traffic.json, so the file of THIS run must be in place before the bench line is taken
for C in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$C -o p -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > $OUT/pmc_$C.json 2> $OUT/pmc_$C.log || exit 1
done
python3 tools/profile_extract.py traffic $(find $OUT/pmc_WRITE_SIZE -name "*.db" | head -1) $(find $OUT/pmc_FETCH_SIZE -name "*.db" | head -1) $OUT/pmc_traffic.json hsw_expand_kernel 9771679744 $COMMIT
echo "pmc traffic done"
cp $OUT/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || exit 1
echo "bench line done"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 bench.py > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof_kt.log || exit 1
DB=$(find $OUT/kt -name "*.db" | head -1)
python3 tools/profile_extract.py stats_by_grid $DB $OUT/bench_kernel_stats.csv 1048576
python3 tools/profile_extract.py stats_by_grid $DB $OUT/bench_all_kernels_stats.csv 0
echo "kernel trace done"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/sq1 -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq1.json 2> $OUT/sq1.log || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq2 -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sq2.json 2> $OUT/sq2.log || exit 1
python3 tools/profile_extract.py sq $OUT/sq_counters.json $(find $OUT/sq1 -name "*.db" | head -1) $(find $OUT/sq2 -name "*.db" | head -1)
echo "sq counters done"
# the three cell forms of the 4,096-block launch side by side: canonical, Montgomery converted at write-out, Montgomery
# converted at emit time (tools/mont_once.py: three launches each)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/m1 -o s -- python3 tools/mont_once.py > $OUT/m1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/m2 -o s -- python3 tools/mont_once.py > $OUT/m2.log 2>&1 || exit 1
python3 tools/profile_extract.py sq $OUT/sq_counters_montgomery.json $(find $OUT/m1 -name "*.db" | head -1) $(find $OUT/m2 -name "*.db" | head -1)
rocprofv3 --kernel-trace --stats -d $OUT/m3 -o s -- python3 tools/mont_once.py > $OUT/m3.log 2>&1 || exit 1
python3 tools/profile_extract.py stats_by_grid $(find $OUT/m3 -name "*.db" | head -1) $OUT/montgomery_kernel_stats.csv 4096
echo "montgomery counters done"
rm -rf $OUT/kt $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE $OUT/sq1 $OUT/sq2 $OUT/m1 $OUT/m2 $OUT/m3
ls -la $OUT
