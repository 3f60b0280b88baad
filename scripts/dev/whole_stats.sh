ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
flags="--steps 10 --warmup 2 --secondary 0 --full-metrics 0 --cpu-budget 0 --cold 0 --configs 0 --train 0 --projection 0 --whole 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/s3_whole_stats" -- python3 "$ROOT/bench.py" $flags > $ROOT/gpurun_out/s3_whole_bench.json 2>/dev/null
find "$ROOT/gpurun_out/s3_whole_stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/s3_whole_kernel_stats.csv"
rm -rf "$ROOT/gpurun_out/s3_whole_stats"
head -16 $ROOT/gpurun_out/s3_whole_kernel_stats.csv | cut -c1-150
