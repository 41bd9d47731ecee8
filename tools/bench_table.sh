set -e
o=gpurun_out/tbl; mkdir -p $o
b() { n=$1; shift; python bench.py --no-cpu-baseline "$@" > $o/$n.json 2>/dev/null; }
b full_8192 --batch 8192 --steps 200
b full_65536 --batch 65536 --steps 30
b ss3_1024 --workload ss3 --steps 1000
b ss3_8192 --workload ss3 --batch 8192 --steps 200
b mixed_1024 --workload mixed --steps 1000
b mixed_8192 --workload mixed --batch 8192 --steps 200
b red_1024 --workload reduced --steps 500
b red_8192 --workload reduced --batch 8192 --steps 100
b redf32_1024 --workload reduced --dtype f32 --steps 500
b redf32_65536 --workload reduced --dtype f32 --batch 65536 --steps 20
b f32_1024 --dtype f32 --steps 1000
b f32_65536 --dtype f32 --batch 65536 --steps 30
b nohqp_1024 --no-hqp --steps 1000
python tools/pcie_rate.py > $o/pcie.txt 2>&1
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/tbl/*.json")):
    d=json.load(open(f)); print(os.path.basename(f)[:-5], round(d["value"]/1e6,3), "M/s", round(d["ms_per_step"],4), "ms")
print(open("gpurun_out/tbl/pcie.txt").read()[-600:])
PY
