#!/bin/bash
# round 2, GPU call A: stream-copy variants, the whole -m gpu suite (with the new C3/C4/C5 tests), bench, kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r02a
python3 -c "import bench; print('gpu count seen by bench.py:', bench._gpu_count())" > gpurun_out/r02a/gpucount.txt 2>&1
tools/native/copy_bench 2147483648 10 > gpurun_out/r02a/copy_bench.txt 2>&1
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r02a/pytest.txt 2>&1
echo "pytest exit=$?" >> gpurun_out/r02a/pytest.txt
tail -5 gpurun_out/r02a/pytest.txt
timeout -k 10 300 python3 bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
echo "bench exit=$?"
tools/prof.sh r02a
