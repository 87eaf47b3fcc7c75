cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03d; rm -rf $O; mkdir -p $O
python3 $R/tools/shared_overhead.py > $O/shared_overhead.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/shared_overhead.py --reps 3 > $O/trace.txt 2>&1
python3 $R/tools/step_timeline.py $O/trace 2 > $O/timeline.txt 2>&1
