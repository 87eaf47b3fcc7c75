cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_cons
mkdir -p $OUT
CMD="python3 $REPO/tools/cbench.py --reps 2 --paths consensus --measures raw"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $CMD > $OUT/c.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/d -- $CMD > $OUT/d.log 2>&1
python3 - <<PY
import csv,glob
from collections import defaultdict
for d in "abcd":
    for p in glob.glob("$OUT/%s/*/*_counter_collection.csv" % d):
        acc=defaultdict(list)
        for r in csv.DictReader(open(p)):
            if "consensus_pair" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(d, k, sum(v)/len(v))
PY
tail -3 $OUT/d.log
