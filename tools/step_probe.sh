# on the GPU box: rocprofv3 kernel trace of bench.py at C2 and C3raw and one step of each as a timeline (gpurun_out/r03b/*_timeline.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03b; rm -rf $O; mkdir -p $O
for w in C2 C3raw; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extra > $O/$w.json 2> $O/$w.err
echo $w rc=$?
python3 $R/tools/step_timeline.py $O/$w 3 > $O/${w}_timeline.txt 2>&1
done
