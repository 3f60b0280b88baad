R=$GRAFT_REPO_ROOT
for v in 0 1; do cp $R/build/ablate/lib$v.so $R/when-do-gnns-help_amd/lib/libwdg_hip.so; echo "== library variant $v (0: 1024 threads x 1 per CU; 1: 512 threads x 2 per CU)"
python3 $R/scripts/dev/try_two_wg.py 1000 10 1 2 4; python3 $R/scripts/dev/try_two_wg.py 1200 10 1 2; done
