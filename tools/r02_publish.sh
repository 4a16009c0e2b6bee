#!/bin/bash
# Copies the summaries of tools/r02_collect.sh (gpurun_out/r02z/, scratch) into profiles/r02/ under the names DESIGN.md,
# profiles/README.md and the roofline_inputs_*.json `source` strings cite, then regenerates the roofline inputs.
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/r02z
D=$R/profiles/r02
mkdir -p $D
for f in bench_c2.json bench_c4.json bench_sharded_w1.json bench_sharded_w1_c4.json c3_pairs.json c5_loop_300.json c5_closed_loop.json kernel_stats_c2.csv kernel_stats_c4.csv \
         pmc_sq_c2.txt pmc_traffic_c2.txt pmc_traffic_c4.txt; do
  [ -s $S/$f ] && cp $S/$f $D/z_$f
done
python3 $R/tools/make_roofline_inputs.py $S r02
