cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/nrun -- python3 $R/tools/nrun_bench.py --cases ${CASES:-0.2:0.1} --paths auto --reps 3 > $O/nrun.txt 2>&1
echo rc=$?
python3 $R/tools/step_timeline.py $O/nrun 2 > $O/nrun_timeline.txt 2>&1
