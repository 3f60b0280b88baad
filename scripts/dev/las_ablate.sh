# times las_small_fused in the sweep step for the library variants under build/ablate/lib<v>.so (dev only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-0}; do cp $R/build/ablate/lib$v.so $R/when-do-gnns-help_amd/lib/libwdg_hip.so
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$v -- python3 $R/bench.py --steps 50 --warmup 5 --secondary 0 --full-metrics 0 --cold 0 --cpu-budget 0 > /tmp/abl$v.json 2>/dev/null
echo "variant $v: $(grep las_small $(find /tmp/abl$v -name '*kernel_stats.csv' | head -1) | cut -d, -f2-5) $(python3 -c "import json;print(json.load(open('/tmp/abl$v.json'))['ms_per_step'])")"; done
