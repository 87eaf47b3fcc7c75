# on the GPU box: C2 step timeline, then 200,000 x 1,000 jc69 / raw on the consensus path: default build, the -DDST_DBG_ALIGN_JC69 measurement build, PMC passes (profiles/r03/c5_jc69_pmc.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03a; rm -rf $O; mkdir -p $O
if [ -z "$SKIP_C2" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python3 $R/bench.py --workload C2 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/c2.json 2> $O/c2.err
echo c2 rc=$?
python3 $R/tools/step_timeline.py $O/c2 3 > $O/c2_timeline.txt 2>&1
fi
python3 $R/tools/cbench.py --n 200000 --len 1000 --measures raw,jc69 --paths consensus --reps 4 > $O/c5_default.txt 2>&1
echo c5 rc=$?
DST_LIB_PATH=$R/build/variants/libdistance_hip_alignjc69.so python3 $R/tools/cbench.py --n 200000 --len 1000 --measures raw,jc69 --paths consensus --reps 4 > $O/c5_alignjc69.txt 2>&1
echo c5 variant rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/c5_pmc_sq -- python3 $R/tools/cbench.py --n 200000 --len 1000 --measures jc69 --paths consensus --reps 2 > $O/c5_pmc_sq.txt 2>&1
echo pmc rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $O/c5_pmc_lds -- python3 $R/tools/cbench.py --n 200000 --len 1000 --measures jc69 --paths consensus --reps 2 > $O/c5_pmc_lds.txt 2>&1
echo pmc2 rc=$?
