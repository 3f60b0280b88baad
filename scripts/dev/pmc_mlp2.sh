# SQ counters of the fused transform alone (dev tool; two passes, --kernel-trace only beside --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export WDG_MLP2_SPLIT=${1:-1}
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pm1 -- python3 $R/scripts/dev/time_mlp2.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d /tmp/pm2 -- python3 $R/scripts/dev/time_mlp2.py > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_IFETCH SQ_ACTIVE_INST_FLAT --output-format csv -d /tmp/pm3 -- python3 $R/scripts/dev/time_mlp2.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ("/tmp/pm1","/tmp/pm2","/tmp/pm3"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "mlp2_split" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k in sorted(acc): print(k, acc[k]/max(n[k],1)*1.0, "x", n[k])
PY
