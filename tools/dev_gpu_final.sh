#!/bin/bash
# round-end GPU call: full validation, rocprofv3 stats + PMC passes of the bench command, the stage timeline of the timed twin (VT),
# the table of secondary rates.  Everything lands under gpurun_out/final/.
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/final; mkdir -p $O
OUT=final bash tools/dev_gpu_full.sh > $O/full.txt 2>&1; tail -14 $O/full.txt
bash tools/profile.sh final_prof full > $O/profile.log 2>&1; cd $R
python tools/summarize_profile.py gpurun_out/final_prof $O/r04c > $O/summarize.log 2>&1; tail -3 $O/summarize.log
# (the timed twin: make -C libdwbc_amd/csrc experiment VARIANT=<VT> XFLAGS="-DDWBC_STAGE_TIMERS -DDWBC_PMASK=0x3df7ull")
if [ -f libdwbc_amd/libdwbc_hip_${VT:-w9t}.so ]; then DWBC_LIB_VARIANT=${VT:-w9t} timeout -k 10 300 python tools/stage_times_pair.py > $O/timeline.txt 2>&1; grep -v amdgpu.ids $O/timeline.txt | head -40; fi
timeout -k 10 900 bash tools/bench_table.sh > $O/bench_table.txt 2>&1; tail -22 $O/bench_table.txt
timeout -k 10 300 python tools/gc_rate.py > $O/gc_rate.txt 2>&1; grep -v amdgpu $O/gc_rate.txt
STRESS_CFGS=gc_feet_lhand,gc_feet_rhand,gc_foot_hands,gc_any STRESS_SEEDS=2 timeout -k 10 600 python tools/stress_parity.py > $O/gc_parity.txt 2>&1; grep -v amdgpu $O/gc_parity.txt | tail -5
