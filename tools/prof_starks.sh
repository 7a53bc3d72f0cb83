#!/bin/bash
# rocprofv3 kernel trace + statistics of two zkEVM-size proofs (bench_starks.py): the source of bench.py's genproof.kernel_rooflines and of
# tools/copybuffer_where.py.  Run on the GPU box from the repo root: gpurun -- 'bash tools/prof_starks.sh r05'
set -u
TAG=${1:-r05}
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
OUT=$ROOT/gpurun_out/prof_starks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT -o starks --output-format csv -- python3 $ROOT/bench_starks.py --proofs 2 --check-rows 0 > $OUT/bench_starks.json 2> $OUT/bench_starks.err
echo "rc=$?" > $OUT/status.txt
cd $ROOT
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
python3 tools/copybuffer_where.py $OUT $OUT/copybuffer_where.json > $OUT/copybuffer_where.txt 2>&1
# the trace itself is large: keep the summaries only
find $OUT -name '*kernel_trace.csv' -size +20M -delete
tail -3 $OUT/copybuffer_where.txt
