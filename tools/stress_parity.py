"""One-off randomized parity sweep on the GPU: the fp64 product kernel against oracle/dwbc_oracle.c on 5 seeds x 4096
instances for each contact / task configuration of the test-suite (result kept in profiles/r01_final_parity_stress.txt)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import libdwbc_amd as D
from tests import cases
from oracle import orc
M = orc.make_model(cases.tocabi_model())
def make(B, tasks):
    model = D.Model.from_urdf(cases.URDF); w = D.Batch(model, B, device=0)
    for c in cases.CONTACTS_2: w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links: w.add_task(lv, mode, link, pt)
    w.set_torque_limit(np.array(cases.TAU_LIM)); return w
NS = int(os.environ.get("STRESS_SEEDS", "5"))  # seeds per configuration (STRESS_SEEDS=20 for a long soak)
B = 4096
cfgs = {"ds": (cases.TASKS_2LEVEL, {}), "ds_yaw": (cases.TASKS_2LEVEL, dict(yaw=True)), "mixed": (cases.TASKS_2LEVEL, dict(contact_mode="mixed")),
        "ss_L": (cases.TASKS_3LEVEL_SWING_R, dict(contact_mode="L", levels=3)), "ss_R": (cases.TASKS_3LEVEL_SWING_L, dict(contact_mode="R", levels=3))}
ONLY = os.environ.get("STRESS_CFGS")  # comma-separated subset (development builds that carry the two-level kernels only: ds,ds_yaw,mixed)
for name, (tasks, kw) in cfgs.items():
    if ONLY and name not in ONLY.split(","): continue
    w = make(B, tasks); S = orc.make_setup(cases.CONTACTS_2, tasks, cases.TAU_LIM)
    worst = 0.0; mism = 0; tot = 0; okc = 0
    for seed in range(NS):
        q, fl, fs = cases.synth_batch(B, seed=9000 + seed, **kw)
        w.set_state(q); w.set_contact(fl); w.set_fstar_all(fs); w.solve()
        tau, st = w.get("tau"), w.get("status")
        tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 16)
        mism += int((st != sr).sum()); tot += B
        ok = (st == 1) & (sr == 1); okc += int(ok.sum())
        worst = max(worst, float(np.abs(tau[ok] - tr[ok]).max()))
    print(f"{name:7s} instances {tot}  status mismatches {mism}  ok {okc}  max|tau - oracle| {worst:.3e}", flush=True)

# the product (two-contact) kernels on contact pairs that include a hand: four registered contacts, two active per instance
PAIRS = [[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 1], [1, 0, 0, 1], [0, 1, 1, 0], [1, 1, 0, 0]]
B = 4096
w = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
for c in cases.CONTACTS_4: w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
for lv, links in enumerate(cases.TASKS_2LEVEL):
    for mode, link, pt in links: w.add_task(lv, mode, link, pt)
w.set_torque_limit(np.array(cases.TAU_LIM))
S = orc.make_setup(cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
worst = 0.0; mism = 0; tot = 0; okc = 0
for seed in range(NS):
    q, _, fs = cases.synth_batch(B, seed=9100 + seed, yaw=True)
    rng = np.random.default_rng(seed)
    fl = np.array([PAIRS[i] for i in rng.integers(0, len(PAIRS), B)], np.uint8)
    w.set_state(q); w.set_contact(fl); w.set_fstar_all(fs); w.solve()
    tau, st = w.get("tau"), w.get("status")
    tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 16)
    mism += int((st != sr).sum()); tot += B
    ok = (st == 1) & (sr == 1); okc += int(ok.sum())
    worst = max(worst, float(np.abs(tau[ok] - tr[ok]).max()))
print(f"hand_pairs ({w.kernel_name()[:40]}) instances {tot}  status mismatches {mism}  ok {okc}  max|tau - oracle| {worst:.3e}", flush=True)

# three simultaneously active contacts through the general-contact kernel (dwbc_cycle_gc.h): random contact sets per instance
def make_gc(B, tasks):
    w = D.Batch(D.Model.from_urdf(cases.URDF), B, device=0)
    for c in cases.CONTACTS_4: w.add_contact(c["link"], c["point"], c["lx"], c["ly"], c["mu"], c["muz"])
    for lv, links in enumerate(tasks):
        for mode, link, pt in links: w.add_task(lv, mode, link, pt)
    w.set_torque_limit(np.array(cases.TAU_LIM)); w.set_max_active_contacts(3); return w
SETS = {"gc_feet_lhand": [[1, 1, 1, 0]], "gc_feet_rhand": [[1, 1, 0, 1]], "gc_foot_hands": [[1, 0, 1, 1], [0, 1, 1, 1]],
        "gc_any": [[1, 1, 1, 0], [1, 1, 0, 1], [1, 0, 1, 1], [0, 1, 1, 1], [1, 1, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 1, 0], [0, 1, 0, 1]]}
B = 2048
for name, sets in SETS.items():
    w = make_gc(B, cases.TASKS_2LEVEL); S = orc.make_setup(cases.CONTACTS_4, cases.TASKS_2LEVEL, cases.TAU_LIM)
    worst = 0.0; worstw = 0.0; mism = 0; tot = 0; okc = 0
    for seed in range(max(3, NS * 3 // 5)):
        q, _, fs = cases.synth_batch(B, seed=9100 + seed, yaw=True)
        rng = np.random.default_rng(seed)
        fl = np.array([sets[i] for i in rng.integers(0, len(sets), B)], np.uint8)
        w.set_state(q); w.set_contact(fl); w.set_fstar_all(fs); w.solve()
        tau, wrn, st = w.get("tau"), w.get("wrench"), w.get("status")
        tr, wr, sr, _ = orc.cycle_batch(M, S, q, fl, fs, 16)
        mism += int((st != sr).sum()); tot += B
        ok = (st == 1) & (sr == 1); okc += int(ok.sum())
        worst = max(worst, float(np.abs(tau[ok] - tr[ok]).max()))
        worstw = max(worstw, float(np.abs(wrn[ok] - wr[ok][:, :18]).max()))
    print(f"{name:14s} instances {tot}  status mismatches {mism}  ok {okc}  max|tau - oracle| {worst:.3e}  max|wrench - oracle| {worstw:.3e}", flush=True)
