# on the GPU box: SQ counters of the pack at the default workload and at C2 (one pass; gpurun_out/r03pack)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03pack; rm -rf $O; mkdir -p $O
for w in C3raw; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/$w -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/$w.json 2> $O/$w.err
echo $w rc=$?
done
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "pack" in k or "slot_fill" in k or "site_bucket" in k:
            print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
