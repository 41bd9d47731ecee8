#!/bin/bash
# full validation GPU call: the GPU test-suite, smoke, bench (driver flags + 1000 steps), pair parity sweep
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/${OUT:-full}; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>&1
timeout -k 10 300 python bench.py --steps 1000 --no-cpu-baseline > $O/bench_1000.json 2>&1
timeout -k 10 300 python bench.py --steps 200 --batch 8192 --no-cpu-baseline > $O/bench_8192.json 2>&1
STRESS_SEEDS=4 timeout -k 10 600 python tools/stress_parity_pair.py > $O/parity_pair.txt 2>&1
python - <<PY
import json
for n in ("steps20","1000","8192"):
    try:
        l=[x for x in open(f"$O/bench_{n}.json") if x.startswith("{")][-1]; d=json.loads(l)
        print(n, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["status_ok_fraction"], d["roofline"]["kernel"][:50])
    except Exception as e: print(n, "failed", e)
PY
grep -v amdgpu.ids $O/parity_pair.txt | tail -6
