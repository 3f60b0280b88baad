cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pg1 -- python3 $R/scripts/dev/time_gram.py 55 2000 500 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d /tmp/pg2 -- python3 $R/scripts/dev/time_gram.py 55 2000 500 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ("/tmp/pg1","/tmp/pg2"):
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "gram_map" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k in sorted(acc): print(k, acc[k]/max(n[k],1)*1.0, "x", n[k])
PY
