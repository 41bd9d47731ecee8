#!/bin/bash
# development GPU call: bench + parity of an experiment build (V) and the stage timeline of its timed twin (VT)
R=$GRAFT_REPO_ROOT; cd $R; V=${V:-x2}; VT=${VT:-x2t}; O=gpurun_out/${OUT:-r2}; mkdir -p $O
DWBC_LIB_VARIANT=$V timeout -k 10 300 python bench.py --steps 1000 --no-cpu-baseline > $O/bench_$V.json 2>&1 && \
DWBC_LIB_VARIANT=$V timeout -k 10 300 python bench.py --steps 200 --batch 8192 --no-cpu-baseline > $O/bench_${V}_8192.json 2>&1 && \
DWBC_LIB_VARIANT=$VT timeout -k 10 300 python tools/stage_times_pair.py > $O/timeline_$VT.txt 2>&1 && \
DWBC_LIB_VARIANT=$V STRESS_SEEDS=${SEEDS:-4} timeout -k 10 600 python tools/stress_parity_pair.py > $O/parity_$V.txt 2>&1 && \
DWBC_LIB_VARIANT=$V timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "double_support_batch_vs_oracle or golden" > $O/pytest_$V.txt 2>&1
tail -3 $O/pytest_$V.txt
python - <<PY
import json
for n in ("$V","${V}_8192"):
    try:
        l=[x for x in open(f"$O/bench_{n}.json") if x.startswith("{")][-1]; d=json.loads(l)
        print(n, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["config"]["status_ok_fraction"], d["roofline"]["kernel"][:60])
    except Exception as e: print(n, "failed", e)
PY
cat $O/timeline_$VT.txt | tail -40
cat $O/parity_$V.txt | tail -9
