#!/bin/bash
# round 3: the measurements DESIGN.md / bench.py quote, in one GPU call -> gpurun_out/r03z/   (tools/r03_publish.sh copies them to profiles/r03/)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r03z
mkdir -p $O
PART=${1:-all}   # a | b | c | all (separate GPU calls keep each well under the time limit of one); d = the 2 000-sweep soak, on its own
if [ "$PART" = "a" ] || [ "$PART" = "all" ]; then
# 1. bench (C2) + kernel trace of the same timed region
timeout -k 10 300 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tools/prof.sh r03z > $O/prof_c2.txt 2>&1
cp gpurun_out/prof_r03z/r03z_kernel_stats.csv $O/kernel_stats_c2.csv
# 2. PMC: HBM traffic of k_match2 (separate passes), then the SQ picture
tools/pmc2.sh r03z_f k_match2 "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_c2.txt 2>&1
tools/pmc2.sh r03z_s k_match2 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" > $O/pmc_sq_c2.txt 2>&1
# 3. C4: kernel trace and traffic first (bench.py reads the inputs made from them), bench afterwards in a second call of this script's tail
tools/prof.sh r03z4 --scan 500000 --map 20000000 --voxel 0.02 > $O/prof_c4.txt 2>&1
cp gpurun_out/prof_r03z4/r03z4_kernel_stats.csv $O/kernel_stats_c4.csv
PMC_ARGS="--scan 500000 --map 20000000 --voxel 0.02" tools/pmc2.sh r03z4_f k_match2 "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_c4.txt 2>&1
# roofline inputs from the traces just taken, then the benches that quote them
python3 tools/make_roofline_inputs.py $O r03 > $O/roofline_inputs.txt 2>&1
timeout -k 10 300 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 400 python3 bench.py --scan 500000 --map 20000000 --voxel 0.02 --steps 5 --warmup 2 --no-cpu --batch-pairs 0 > $O/bench_c4.json 2> $O/bench_c4.err
fi
if [ "$PART" = "b" ] || [ "$PART" = "all" ]; then
# 4. the first iteration of a call against a converged one (row-disc far search vs the ring search)
python3 tools/first_iter3.py > $O/first_iter_c2.json 2>/dev/null
O3S_FAR=0 python3 tools/first_iter3.py > $O/first_iter_c2_ring.json 2>/dev/null
CFG=c4 python3 tools/first_iter3.py > $O/first_iter_c4.json 2>/dev/null
CFG=c4 O3S_FAR=0 python3 tools/first_iter3.py > $O/first_iter_c4_ring.json 2>/dev/null
# 5. sharded mode at world size 1 (RCCL in the loop, chain + collectives replayed from one hipGraph) against the unsharded chain
timeout -k 10 200 python3 bench.py --mode sharded --exchange rccl --no-cpu > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
timeout -k 10 300 python3 bench.py --mode sharded --exchange rccl --no-cpu --scan 500000 --map 20000000 --voxel 0.02 --steps 5 --warmup 2 > $O/bench_sharded_w1_c4.json 2> $O/bench_sharded_w1_c4.err
tools/prof_any.sh r03zs --mode sharded --exchange rccl --no-cpu --steps 6 --warmup 2 > $O/prof_sharded_w1.txt 2>&1
# 6. config 3 on one GPU (64 pairs) and 8 pairs per GPU through bench.py, config 5 loops
timeout -k 10 300 python3 tools/c3_pairs.py --out $O/c3_pairs.json > /dev/null 2> $O/c3.err
timeout -k 10 300 python3 bench.py --pairs-per-gpu 8 --steps 5 --no-cpu --batch-pairs 0 > $O/bench_pairs8.json 2> $O/bench_pairs8.err
LIDAR=1 SCANS=300 STEP=0.25 NORMALS=1 GEN_PROCS=12 timeout -k 10 300 python3 tools/mapping_loop.py > $O/c5_loop_300.json 2> $O/c5.err
fi
if [ "$PART" = "c" ] || [ "$PART" = "all" ]; then
# 6b. config 5 through the compiled driver: one thread; sweep staged by a second thread; sweep pre-processed by it; the same with
#     the sort-based insert; the closed loop with submaps and loop closures
SCANS=300 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper.json 2> $O/c5_compiled.err
SCANS=300 PREFETCH=1 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_prefetch.json 2> $O/c5_compiled_prefetch.err
SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed.json 2> $O/c5_compiled_preprocessed.err
O3S_INSERT_SORT=1 SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed_sort_insert.json 2> $O/c5_compiled_preprocessed_sort.err
LOOP=1 SCANS=640 SUBMAP_RADIUS=20 PREFETCH=2 PRELOAD=1 timeout -k 10 400 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_closed_loop.json 2> $O/c5_compiled_closed_loop.err
ASYNC_CLOSURES=1 LOOP=1 SCANS=640 SUBMAP_RADIUS=20 PREFETCH=2 PRELOAD=1 timeout -k 10 400 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_closed_loop_async_closures.json 2> $O/c5_compiled_closed_loop_async.err
# sweeps WITHOUT normals (what a lidar driver delivers): estimated on the device (radius 1 m, knn 10) inside the pre-processing
ESTIMATE_NORMALS=1.0,10 SCANS=300 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_estimated_normals.json 2> $O/c5_compiled_en.err
ESTIMATE_NORMALS=1.0,10 PINNED=1 SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed_estimated_normals.json 2> $O/c5_compiled_en2.err
# the reference's tutorial setting: the ICP reference renewed every 2 s (every 20th sweep), sweeps in page-locked memory
REF_PERIOD=2.0 PINNED=1 SCANS=300 PREFETCH=2 PRELOAD=1 timeout -k 10 300 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_mapper_preprocessed_ref2s.json 2> $O/c5_compiled_preprocessed_ref2s.err
# 7. per-scan loop timeline (compiled driver, sweeps pre-processed by the receiving thread): busy fraction and the gaps
SCANS=120 PREFETCH=2 PRELOAD=1 tools/prof_mapper_cpp.sh r03zw
python3 tools/loop_gaps.py gpurun_out/prof_r03zw/r03zw_kernel_trace.csv > $O/w_loop_gaps.txt
cp gpurun_out/prof_r03zw/r03zw_kernel_stats.csv $O/w_loop_kernel_stats.csv
# ... and of the one-thread Python loop, for the host round trips it shows
LIDAR=1 SCANS=120 STEP=0.25 NORMALS=1 GEN_PROCS=12 CPU_SCANS=0 tools/prof_loop.sh r03zp
python3 tools/loop_gaps.py gpurun_out/prof_r03zp/r03zp_kernel_trace.csv > $O/w_loop_gaps_python_one_thread.txt
fi
if [ "$PART" = "d" ]; then
# 8. soak: 2 000 sweeps (500 m, three laps), 20 m submaps, every loop closure refined inline   -> copy to profiles/r03/z_c5_compiled_closed_loop_2000.json
LOOP=1 SCANS=2000 SUBMAP_RADIUS=20 PREFETCH=2 PRELOAD=1 GEN_PROCS=14 timeout -k 10 1000 python3 tools/mapper_cpp_bench.py > $O/c5_compiled_closed_loop_2000.json 2> $O/c5_compiled_closed_loop_2000.err
fi
echo done
