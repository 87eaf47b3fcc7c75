# on the GPU box: the shared preparation beside the replicated one (one-rank RCCL communicator), with a kernel timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03shared; rm -rf $O; mkdir -p $O
python3 $R/tools/shared_overhead.py > $O/overhead.txt 2> $O/overhead.err; echo rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/shared_overhead.py --reps 4 > $O/trace.txt 2> $O/trace.err; echo rc=$?
python3 $R/tools/step_timeline.py $O/trace 1 > $O/timeline.txt 2>&1
cat $O/overhead.txt
