#!/bin/bash
# round 5: the measurements DESIGN.md / bench.py quote -> gpurun_out/r05z/ (copy what is quoted into profiles/r05/ afterwards).
# usage (GPU box): tools/r05_collect.sh a|b|c   (separate GPU calls keep each well under the time limit of one)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05z
mkdir -p $O
PART=${1:-a}
if [ "$PART" = "a" ]; then
# 1. kernel trace of the bench's timed region (C2, C4), PMC traffic of k_match2 in separate passes, SQ picture, resource table
tools/prof_any.sh r05z --timing-only --steps 8 --warmup 2 > $O/prof_c2.txt 2>&1
cp gpurun_out/prof_r05z/r05z_kernel_stats.csv $O/kernel_stats_c2.csv
tools/pmc2.sh r05z_f k_match2 "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_c2.txt 2>&1
tools/pmc2.sh r05z_s k_match2 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" > $O/pmc_sq_c2.txt 2>&1
tools/prof_any.sh r05z4 --timing-only --steps 4 --warmup 2 --scan 500000 --map 20000000 --voxel 0.02 > $O/prof_c4.txt 2>&1
cp gpurun_out/prof_r05z4/r05z4_kernel_stats.csv $O/kernel_stats_c4.csv
PMC_ARGS="--scan 500000 --map 20000000 --voxel 0.02" tools/pmc2.sh r05z4_f k_match2 "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_c4.txt 2>&1
python3 tools/make_roofline_inputs.py $O r05 > $O/roofline_inputs.txt 2>&1
make -C open3d_slam_advanced_rss_2024_public_amd/csrc resource-usage 2>&1 | python3 tools/resource_table.py > $O/kernel_resources.txt
# 2. the driver's command, with the roofline inputs just made in place
timeout -k 10 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
fi
if [ "$PART" = "b" ]; then
# 3. first iteration / converged, far-search split (hooks build), yaml-chain kernel timeline, host-buffer calls
python3 tools/first_iter3.py > $O/first_iter_c2.json 2>/dev/null
CFG=c4 STATS=1 python3 tools/first_iter3.py > $O/first_iter_c4.json 2>/dev/null
O3S_LIB_VARIANT=hooks python3 tools/far_split.py > $O/far_split_c2.json 2>/dev/null
CFG=c4 O3S_LIB_VARIANT=hooks python3 tools/far_split.py > $O/far_split_c4.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05z_yaml_trace -- python3 $R/tools/r04_yaml_trace.py > $R/$O/yaml_trace.log 2>&1)
python3 tools/r04_trace_summary.py gpurun_out/r05z_yaml_trace > $O/yaml_trace_summary.txt 2>&1
python3 tools/r04_hostbuf.py > $O/host_buffers.json 2>/dev/null
# 4. sharded mode at world size 1 (RCCL in the loop) against the unsharded chain; 8 pairs in flight; config 3
timeout -k 10 200 python3 bench.py --mode sharded --exchange rccl --no-cpu > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
timeout -k 10 300 python3 bench.py --mode sharded --exchange rccl --no-cpu --scan 500000 --map 20000000 --voxel 0.02 --steps 5 --warmup 2 > $O/bench_sharded_w1_c4.json 2> $O/bench_sharded_w1_c4.err
tools/prof_any.sh r05zs --mode sharded --exchange rccl --no-cpu --steps 6 --warmup 2 > $O/prof_sharded_w1.txt 2>&1
cp gpurun_out/prof_r05zs/r05zs_kernel_stats.csv $O/kernel_stats_sharded_w1.csv
timeout -k 10 300 python3 bench.py --pairs-per-gpu 8 --steps 5 --no-cpu --batch-pairs 0 --no-c4 > $O/bench_pairs8.json 2> $O/bench_pairs8.err
timeout -k 10 300 python3 tools/c3_pairs.py --out $O/c3_pairs.json > /dev/null 2> $O/c3.err
fi
if [ "$PART" = "c" ]; then
# 5. config 5 through the compiled driver (sweep pre-processed by the receiving thread), closed loop, timeline
SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed.json 2> $O/c5_a.err
PINNED=1 SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed_pinned.json 2> $O/c5_b.err
SCANS=300 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper.json 2> $O/c5_c.err
LOOP=1 SCANS=640 SUBMAP_RADIUS=20 PREFETCH=2 PRELOAD=1 timeout -k 10 400 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_closed_loop.json 2> $O/c5_d.err
REF_PERIOD=2.0 SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed_ref2s.json 2> $O/c5_e.err
REF_PERIOD=2.0 PINNED=1 SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed_ref2s_pinned.json 2> $O/c5_f.err
ALSO_REF_PERIOD=2.0 PINNED=1 SCANS=300 PREFETCH=3 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_three_stages_pinned.json 2> $O/c5_g.err
LOOP=1 SCANS=2000 STEP=0.25 SUBMAP_RADIUS=20 PREFETCH=2 PRELOAD=1 timeout -k 10 600 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_closed_loop_2000.json 2> $O/c5_h.err
SCANS=120 PREFETCH=2 PRELOAD=1 tools/prof_mapper_cpp.sh r05zw
python3 tools/loop_gaps.py gpurun_out/prof_r05zw/r05zw_kernel_trace.csv > $O/w_loop_gaps.txt
cp gpurun_out/prof_r05zw/r05zw_kernel_stats.csv $O/w_loop_kernel_stats.csv
fi
echo done
