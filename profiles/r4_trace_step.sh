cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_trace2; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-context > $O/bench.json 2> $O/bench.err
python $R/profiles/step_timeline.py $(ls $O/trace/*kernel_trace.csv | head -1) 3 > $O/step_timeline.txt 2>&1
sed -n '/composite_fwd/,$p' $O/step_timeline.txt
rm -rf $O/trace
