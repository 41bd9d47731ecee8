"""ctypes loader of the CPU emulation of the kernel source (tests/emu/emu_cycle.cpp).  TEST HARNESS ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def lib(f32=False):
    if f32 not in _libs:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = C.CDLL(os.path.join(_HERE, "libdwbc_emu_f32.so" if f32 else "libdwbc_emu.so"))
        L.emu_create.restype = C.c_void_p
        L.emu_create.argtypes = [C.c_char_p]
        L.emu_error.restype = C.c_char_p
        L.emu_error.argtypes = [C.c_void_p]
        for f in ("emu_destroy", "emu_nb", "emu_ndof", "emu_fstar_total", "emu_dump_total"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.emu_link_id.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_get_model.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.emu_add_contact.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]
        L.emu_add_task.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.emu_set_tau_lim.argtypes = [C.c_void_p, C.c_void_p]
        L.emu_dump_offset.argtypes = [C.c_void_p, C.c_char_p]
        L.emu_run.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.emu_run_reduced.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.emu_run_other.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        _libs[f32] = L
    return _libs[f32]


class Emu:
    def __init__(self, urdf, contacts, tasks, tau_lim=None, f32=False):
        L = lib(f32)
        self.L = L
        if isinstance(urdf, dict):  # a model given as arrays (the result of model surgery)
            L.emu_create_from_arrays.restype = C.c_void_p
            L.emu_create_from_arrays.argtypes = [C.c_int] + [C.c_void_p] * 7
            arrs = [np.ascontiguousarray(urdf["parent"], np.int32)] + [np.ascontiguousarray(urdf[k], np.float64) for k in ("R_T", "p_T", "axis", "mass", "com", "inertia")]
            self.h = L.emu_create_from_arrays(int(urdf["nb"]), *[a.ctypes.data for a in arrs])
        else:
            self.h = L.emu_create(urdf.encode())
        err = L.emu_error(self.h).decode()
        if err:
            raise RuntimeError(err)
        self.n = L.emu_ndof(self.h)
        self.nb = L.emu_nb(self.h)
        self.m = self.n - 6
        for c in contacts:
            pt = np.asarray(c["point"], dtype=np.float64)
            assert L.emu_add_contact(self.h, c["link"], pt.ctypes.data, c["lx"], c["ly"], c.get("mu", 0.2), c.get("muz", 0.2)) >= 0
        for lv, links in enumerate(tasks):
            if isinstance(links, int):  # TASK_CUSTOM level of that many dof
                assert L.emu_add_custom_task(C.c_void_p(self.h), lv, links) == 1, L.emu_error(self.h)
                continue
            for mode, link, pt in links:
                p = np.asarray(pt, dtype=np.float64)
                assert L.emu_add_task(self.h, lv, mode, link, p.ctypes.data) == 1, L.emu_error(self.h)
        if tau_lim is not None:
            t = np.asarray(tau_lim, dtype=np.float64)
            L.emu_set_tau_lim(self.h, t.ctypes.data)
        self.ncon = len(contacts)
        self.F = L.emu_fstar_total(self.h)
        self.D = L.emu_dump_total(self.h)

    def model_arrays(self):
        nb = self.nb
        out = dict(parent=np.zeros(nb, np.int32), R_T=np.zeros((nb, 3, 3)), p_T=np.zeros((nb, 3)), axis=np.zeros((nb, 3)),
                   mass=np.zeros(nb), com=np.zeros((nb, 3)), inertia=np.zeros((nb, 3, 3)))
        self.L.emu_get_model(self.h, *[out[k].ctypes.data for k in ("parent", "R_T", "p_T", "axis", "mass", "com", "inertia")])
        return out

    def set_traj(self, level, link_index, slot, gains15):
        g = np.ascontiguousarray(gains15, np.float64)
        self.L.emu_set_traj(C.c_void_p(self.h), level, link_index, slot, C.c_void_p(g.ctypes.data))

    def run(self, q, flags, fstar, dump=False, reduced=False, qdot=None, traj=None, ctime=None, custom_J=None, hqp=True, dense=False, warm_diag=None, compact=False):
        B = q.shape[0]
        q = np.ascontiguousarray(q, np.float64)
        flags = np.ascontiguousarray(flags, np.uint8)
        fstar = np.ascontiguousarray(fstar, np.float64)
        assert flags.shape == (B, self.ncon) and fstar.shape == (B, self.F)
        tau = np.zeros((B, 3, self.m))
        wr = np.zeros((B, 12))
        st = np.zeros(B, np.int32)
        diag = np.zeros((B, 90), np.int32)
        if warm_diag is not None:  # init = false: the working sets of a previous run (its "diag") seed the QPs
            diag[:] = warm_diag
        self.L.emu_set_warm(1 if warm_diag is not None else 0)
        dmp = np.zeros((B, self.D)) if dump else None
        qd = None if qdot is None else np.ascontiguousarray(qdot, np.float64)
        self.L.emu_set_qdot(C.c_void_p(qd.ctypes.data if qd is not None else None))
        self.L.emu_set_hqp(1 if hqp else 0)
        # compact: True = lean build on the compact LDS map (Lds3); "pair" = the two-wave kernel of dwbc_cycle2p.h, roles run in turn;
        # LDS NaN-poisoned per instance in both
        self.L.emu_set_compact(2 if compact == "pair" else (1 if compact else 0))
        self.L.emu_set_dense(1 if dense else 0)  # two-level runs: TopoGeneric instantiation (dense A^-1 sweep)
        cj = None if custom_J is None else np.ascontiguousarray(custom_J, np.float64)  # (B, n_custom, 6, n)
        self.L.emu_set_custom(C.c_void_p(cj.ctypes.data if cj is not None else None))
        tr = None if traj is None else np.ascontiguousarray(traj, np.float64)
        ct = None if ctime is None else np.ascontiguousarray(ctime, np.float64)
        self.L.emu_set_traj_data(C.c_void_p(tr.ctypes.data if tr is not None else None), C.c_void_p(ct.ctypes.data if ct is not None else None))
        other = (self.n, self.nb) != (39, 34)  # the kernel-pack sizes: emu_run_other
        ok = (self.L.emu_run_other if other else self.L.emu_run_reduced if reduced else self.L.emu_run)(self.h, B, q.ctypes.data, flags.ctypes.data, fstar.ctypes.data, tau.ctypes.data, wr.ctypes.data,
                            st.ctypes.data, diag.ctypes.data, dmp.ctypes.data if dump else None)
        assert ok == 1, self.L.emu_error(self.h)
        return dict(tau=tau, wrench=wr, status=st, diag=diag, dump=dmp)

    def run_gc(self, q, flags, fstar):
        """the general-contact kernel (dwbc_cycle_gc.h, up to three simultaneously active contacts); wrench is (B, 18)"""
        B = q.shape[0]
        q = np.ascontiguousarray(q, np.float64)
        flags = np.ascontiguousarray(flags, np.uint8)
        fstar = np.ascontiguousarray(fstar, np.float64)
        assert flags.shape == (B, self.ncon) and fstar.shape == (B, self.F)
        tau = np.zeros((B, 3, self.m))
        wr = np.zeros((B, 18))
        st = np.zeros(B, np.int32)
        diag = np.zeros((B, 90), np.int32)
        self.L.emu_run_gc.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7
        ok = self.L.emu_run_gc(self.h, B, q.ctypes.data, flags.ctypes.data, fstar.ctypes.data, tau.ctypes.data, wr.ctypes.data, st.ctypes.data, diag.ctypes.data)
        assert ok == 1, self.L.emu_error(self.h)
        return dict(tau=tau, wrench=wr, status=st, diag=diag)

    def dump_field(self, dmp, name, shape):
        off = self.L.emu_dump_offset(self.h, name.encode())
        assert off >= 0
        n = int(np.prod(shape))
        return dmp[:, off : off + n].reshape((dmp.shape[0],) + tuple(shape))


class EmuHQP:
    """host emulation of the batched hierarchical-QP kernels (libdwbc_amd/csrc/dwbc_hqp.h), same entry points as the C-ABI"""

    def __init__(self, B, nv, m, e, has_cost, share_cost=False, solve_first=False):
        L = lib(False)
        self.L = L
        L.emu_hqp_create.restype = C.c_void_p
        L.emu_hqp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.emu_hqp_data.restype = C.POINTER(C.c_double)
        L.emu_hqp_data.argtypes = [C.c_void_p]
        L.emu_hqp_stat.restype = C.POINTER(C.c_int)
        L.emu_hqp_stat.argtypes = [C.c_void_p]
        for f in ("emu_hqp_rec", "emu_hqp_lds_bytes"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p]
        L.emu_hqp_offset.restype = C.c_int
        L.emu_hqp_offset.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.emu_hqp_solve.argtypes = [C.c_void_p]
        L.emu_hqp_destroy.argtypes = [C.c_void_p]
        L.emu_lqp_configure.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.emu_lqp_torque.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.B, self.nv = B, nv
        self.m = np.asarray(m, np.int32)
        self.e = np.asarray(e, np.int32)
        self.hc = np.asarray(has_cost, np.int32)
        self.h = L.emu_hqp_create(B, nv, len(self.m), self.m.ctypes.data, self.e.ctypes.data, self.hc.ctypes.data, 1 if share_cost else 0, 1 if solve_first else 0)
        self.rec = np.ctypeslib.as_array(L.emu_hqp_data(self.h), shape=(B, L.emu_hqp_rec(self.h)))
        self.stat = np.ctypeslib.as_array(L.emu_hqp_stat(self.h), shape=(B, 24))

    def block(self, level, what, shape):
        """view of a per-instance block: what = 0 A 1 a 2 B 3 b 4 H 5 y 6 v 7 w"""
        off = self.L.emu_hqp_offset(self.h, level, what)
        n = int(np.prod(shape))
        return self.rec[:, off : off + n].reshape((self.B,) + tuple(shape))

    def solve(self):
        self.L.emu_hqp_solve(self.h)

    def configure_lqp(self, emu, act, dump, fstar, use_B=False):
        act = np.asarray(act, np.int32)
        dump = np.ascontiguousarray(dump, np.float64)
        fstar = np.ascontiguousarray(fstar, np.float64)
        self.L.emu_lqp_configure(self.h, emu.h, len(act), act.ctypes.data, 1 if use_B else 0, dump.ctypes.data, fstar.ctypes.data)

    def lqp_torque(self, emu, nc, dump, use_B=False):
        dump = np.ascontiguousarray(dump, np.float64)
        tau = np.zeros((self.B, emu.m))
        self.L.emu_lqp_torque(self.h, emu.h, nc, 1 if use_B else 0, dump.ctypes.data, tau.ctypes.data)
        return tau

    def set_exact(self, level, on=True):
        self.L.emu_hqp_set_exact.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.emu_hqp_set_exact(self.h, level, 1 if on else 0)

    def jacc_solve(self, emu, act, level, dump, fstar, prev):
        """CalcSingleTaskTorqueWithJACC_QP for one level; prev = list of (B, rec) arrays of the earlier levels"""
        L = self.L
        L.emu_jacc_rec.restype = C.c_int
        L.emu_jacc_rec.argtypes = [C.c_void_p]
        L.emu_jacc_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rs = L.emu_jacc_rec(emu.h)
        act = np.asarray(act, np.int32)
        dump = np.ascontiguousarray(dump, np.float64)
        fstar = np.ascontiguousarray(fstar, np.float64)
        prev = [np.ascontiguousarray(p, np.float64) for p in prev]
        ptrs = (C.c_void_p * max(1, len(prev)))(*[p.ctypes.data for p in prev])
        out = np.zeros((self.B, rs))
        st = np.zeros(self.B, np.int32)
        L.emu_jacc_solve(self.h, emu.h, len(act), act.ctypes.data, level, dump.ctypes.data, fstar.ctypes.data, ptrs, out.ctypes.data, st.ctypes.data)
        return out, st

    # ---- reduced variants (ConfigureLQP_R, JACC_QP_R and the _NC halves): same device functions on the reduced record
    @staticmethod
    def reduced_record(emu, B, vcd, cd, src, dump):
        L = lib(False)
        L.emu_rrec_total.restype = C.c_int
        L.emu_rrec_total.argtypes = [C.c_int]
        L.emu_reduced_record.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        src = np.asarray(src, np.int32)
        dump = np.ascontiguousarray(dump, np.float64)
        rrec = np.zeros((B, L.emu_rrec_total(vcd + 6)))
        L.emu_reduced_record(emu.h, B, vcd, cd, len(src), src.ctypes.data, dump.ctypes.data, rrec.ctypes.data)
        return rrec

    @staticmethod
    def rrec_field(rrec, RS, name, shape):
        L = lib(False)
        L.emu_rrec_offset.restype = C.c_int
        L.emu_rrec_offset.argtypes = [C.c_int, C.c_char_p]
        off = L.emu_rrec_offset(RS, name.encode())
        n = int(np.prod(shape))
        return rrec[:, off : off + n].reshape((rrec.shape[0],) + tuple(shape))

    def configure_lqp_r(self, emu, act, RS, src, rrec, fstar):
        act = np.asarray(act, np.int32)
        src = np.asarray(src, np.int32)
        fstar = np.ascontiguousarray(fstar, np.float64)
        self.L.emu_lqp_configure_r.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.emu_lqp_configure_r(self.h, emu.h, len(act), act.ctypes.data, RS, len(src), src.ctypes.data, rrec.ctypes.data, fstar.ctypes.data)

    def lqp_torque_r(self, emu, act, RS, rrec):
        act = np.asarray(act, np.int32)
        tau = np.zeros((self.B, RS - 6))
        self.L.emu_lqp_torque_r.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.L.emu_lqp_torque_r(self.h, emu.h, len(act), act.ctypes.data, RS, rrec.ctypes.data, tau.ctypes.data)
        return tau

    def jacc_solve_r(self, emu, act, RS, src, level, rrec, fstar, prev):
        L = self.L
        L.emu_jacc_rec_r.restype = C.c_int
        L.emu_jacc_rec_r.argtypes = [C.c_int]
        L.emu_jacc_solve_r.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        act = np.asarray(act, np.int32)
        src = np.asarray(src, np.int32)
        fstar = np.ascontiguousarray(fstar, np.float64)
        prev = [np.ascontiguousarray(p, np.float64) for p in prev]
        ptrs = (C.c_void_p * max(1, len(prev)))(*[p.ctypes.data for p in prev])
        out = np.zeros((self.B, L.emu_jacc_rec_r(RS)))
        st = np.zeros(self.B, np.int32)
        L.emu_jacc_solve_r(self.h, emu.h, len(act), act.ctypes.data, RS, len(src), src.ctypes.data, level, rrec.ctypes.data, fstar.ctypes.data, ptrs, out.ctypes.data, st.ctypes.data)
        return out, st

    def configure_lqp_nc(self, emu, vcd, level, dump, fstar, prev, prev_off=0):
        """prev: (B, stride) array whose columns prev_off.. hold the reduced answer [base 6 | chain | centroidal 6]"""
        dump = np.ascontiguousarray(dump, np.float64)
        fstar = np.ascontiguousarray(fstar, np.float64)
        prev = np.ascontiguousarray(prev, np.float64)
        self.L.emu_lqp_nc_configure.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self.L.emu_lqp_nc_configure(self.h, emu.h, vcd, level, dump.ctypes.data, fstar.ctypes.data, prev.ctypes.data, prev.shape[1], prev_off)

    def solve_levels(self, n_levels, solve_first):
        self.L.emu_hqp_solve_levels.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.L.emu_hqp_solve_levels(self.h, n_levels, 1 if solve_first else 0)

    def jacc_solve_nc(self, emu, vcd, level, dump, fstar, prev):
        L = self.L
        L.emu_jacc_nc_rec.restype = C.c_int
        L.emu_jacc_nc_rec.argtypes = [C.c_int]
        L.emu_jacc_nc_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        dump = np.ascontiguousarray(dump, np.float64)
        fstar = np.ascontiguousarray(fstar, np.float64)
        prev = np.ascontiguousarray(prev, np.float64)
        out = np.zeros((self.B, L.emu_jacc_nc_rec(emu.n - vcd)))
        st = np.zeros(self.B, np.int32)
        L.emu_jacc_nc_solve(self.h, emu.h, vcd, level, dump.ctypes.data, fstar.ctypes.data, prev.ctypes.data, prev.shape[1], out.ctypes.data, st.ctypes.data)
        return out, st

    def status(self, level):
        return self.stat[:, level]

    def iters(self, level):
        return self.stat[:, 8 + level]

    def null_size(self, level):
        return self.stat[:, 16 + level]

    def __del__(self):
        try:
            self.L.emu_hqp_destroy(self.h)
        except Exception:
            pass
