#!/bin/bash
# round 2: the measurements DESIGN.md / bench.py quote, in one GPU call -> gpurun_out/r02z/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r02z
mkdir -p $O
# 1. bench (C2) + kernel trace of the same command
timeout -k 10 300 python3 bench.py > $O/bench_c2.json 2> $O/bench_c2.err
tools/prof.sh r02z > $O/prof_c2.txt 2>&1
cp gpurun_out/prof_r02z/r02z_kernel_stats.csv $O/kernel_stats_c2.csv
# 2. PMC: HBM traffic of k_match2 (separate passes), then the SQ picture
tools/pmc2.sh r02z_f k_match2 "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_c2.txt 2>&1
tools/pmc2.sh r02z_s k_match2 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" > $O/pmc_sq_c2.txt 2>&1
# 3. C4: bench, kernel trace, traffic
timeout -k 10 400 python3 bench.py --scan 500000 --map 20000000 --voxel 0.02 --steps 5 --warmup 2 --no-cpu --batch-pairs 0 > $O/bench_c4.json 2> $O/bench_c4.err
tools/prof.sh r02z4 --scan 500000 --map 20000000 --voxel 0.02 > $O/prof_c4.txt 2>&1
cp gpurun_out/prof_r02z4/r02z4_kernel_stats.csv $O/kernel_stats_c4.csv
PMC_ARGS="--scan 500000 --map 20000000 --voxel 0.02" tools/pmc2.sh r02z4_f k_match2 "FETCH_SIZE" "WRITE_SIZE" > $O/pmc_traffic_c4.txt 2>&1
# 4. sharded mode at world size 1 (RCCL in the loop) against the unsharded chain
timeout -k 10 200 python3 bench.py --mode sharded --exchange rccl --no-cpu > $O/bench_sharded_w1.json 2> $O/bench_sharded_w1.err
timeout -k 10 300 python3 bench.py --mode sharded --exchange rccl --no-cpu --scan 500000 --map 20000000 --voxel 0.02 --steps 5 --warmup 2 > $O/bench_sharded_w1_c4.json 2> $O/bench_sharded_w1_c4.err
# 5. config 3 on one GPU, config 5 loop
timeout -k 10 300 python3 tools/c3_pairs.py --out $O/c3_pairs.json > /dev/null 2> $O/c3.err
LIDAR=1 SCANS=300 STEP=0.25 NORMALS=1 GEN_PROCS=12 timeout -k 10 300 python3 tools/mapping_loop.py > $O/c5_loop_300.json 2> $O/c5.err
# 6. closed loop with submap switching and loop-closure refinements between resident submaps
LOOP=1 LIDAR=1 SCANS=640 STEP=0.25 NORMALS=1 GEN_PROCS=12 CPU_SCANS=0 SUBMAP_RADIUS=20 timeout -k 10 400 python3 tools/mapping_loop.py > $O/c5_closed_loop.json 2> $O/c5_closed_loop.err
echo done
