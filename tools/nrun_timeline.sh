# on the GPU box: one step of the N-run case (tools/nrun_bench.py, CASES=share:fraction, N records) as a kernel timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/nrun_tl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/nrun_bench.py --n ${N:-50000} --cases ${CASES:-0.05:0.5} --paths auto --reps 3 > $O/out.txt 2> $O/err.txt
echo rc=$?
python3 $R/tools/step_timeline.py $O/trace 1 > $O/timeline.txt 2>&1
cat $O/timeline.txt
