# rocprofv3 kernel trace of the bench's training leg -> gpurun_out/prof_train/ + a text summary
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $R/gpurun_out/prof_train.log 2>&1
tail -1 $R/gpurun_out/prof_train.log | cut -c1-400
python $R/tools/rocprof_summary.py $R/gpurun_out/prof_train > $R/gpurun_out/prof_train_summary.txt 2>&1
head -50 $R/gpurun_out/prof_train_summary.txt | cut -c1-180
