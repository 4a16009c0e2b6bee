#!/bin/bash
# Copies the summaries of tools/r03_collect.sh (gpurun_out/r03z/, scratch) into profiles/r03/ under the names DESIGN.md,
# profiles/README.md and the roofline_inputs_*.json `source` strings cite, then regenerates the roofline inputs.
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/r03z
D=$R/profiles/r03
mkdir -p $D
for f in bench_c2.json bench_c4.json bench_sharded_w1.json bench_sharded_w1_c4.json bench_pairs8.json c3_pairs.json c5_loop_300.json c5_compiled_mapper.json \
         c5_compiled_mapper_prefetch.json c5_compiled_mapper_preprocessed.json c5_compiled_mapper_preprocessed_sort_insert.json c5_compiled_mapper_preprocessed_ref2s.json c5_compiled_mapper_estimated_normals.json c5_compiled_mapper_preprocessed_estimated_normals.json c5_compiled_closed_loop.json c5_compiled_closed_loop_async_closures.json c5_compiled_closed_loop_2000.json kernel_stats_c2.csv kernel_stats_c4.csv pmc_sq_c2.txt pmc_traffic_c2.txt pmc_traffic_c4.txt \
         first_iter_c2.json first_iter_c2_ring.json first_iter_c4.json first_iter_c4_ring.json prof_sharded_w1.txt w_loop_gaps.txt w_loop_gaps_python_one_thread.txt w_loop_kernel_stats.csv; do
  [ -s $S/$f ] && cp $S/$f $D/z_$f
done
python3 $R/tools/make_roofline_inputs.py $S r03
