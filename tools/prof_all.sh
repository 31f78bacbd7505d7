#!/bin/bash
# Everything profiles/<tag>_* is made from, on the GPU box, summarised there (the raw rocprofv3 output is too large to travel back):
#   tools/prof_all.sh <tag>   ->  gpurun_out/profiles_<tag>/   (copy into profiles/)
tag=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
bash $R/tools/prof_bench.sh $tag > $O/prof_bench.log 2>&1 || { tail -5 $O/prof_bench.log; exit 1; }
python3 $R/tools/summarize_prof.py $R/gpurun_out/prof_$tag $O/${tag}_summary.json > /dev/null
cp $(find $R/gpurun_out/prof_$tag/trace -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
grep -h "^{\"metric\"" $R/gpurun_out/prof_$tag/trace.log | tail -1 > $O/${tag}_bench_under_rocprof.json
rm -rf $R/gpurun_out/prof_$tag
bash $R/tools/prof_configs.sh $tag > $O/prof_configs.log 2>&1 || { tail -5 $O/prof_configs.log; exit 1; }
python3 $R/tools/summarize_prof.py $R/gpurun_out/prof_${tag}_frame_constraints $O/${tag}_frame_constraints_summary.json > /dev/null
for cfg in graph optimizer frame_constraints; do
  cp $(find $R/gpurun_out/prof_${tag}_$cfg/trace -name '*kernel_stats.csv' | head -1) $O/${tag}_${cfg}_kernel_stats.csv
  cp $R/gpurun_out/prof_${tag}_$cfg/bench.json $O/${tag}_${cfg}_bench_under_rocprof.json
  rm -rf $R/gpurun_out/prof_${tag}_$cfg
done
cd $R && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${tag}_bench.json 2> $O/bench.err
ls -la $O
