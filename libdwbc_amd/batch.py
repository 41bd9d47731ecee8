"""Host-side mirror of libdwbc's RobotData call sequence for a batch of B robots (reference include/dwbc.h).

    model = Model.from_urdf(path)                      # RobotData::LoadModelData
    wbc = Batch(model, B, device=0)
    wbc.add_contact(link, point, lx, ly)               # AddContactConstraint
    wbc.add_task(level, mode, link)                    # AddTaskSpace
    wbc.set_torque_limit(lim)                          # SetTorqueLimit
    wbc.set_state(q); wbc.set_contact(flags); wbc.set_fstar(level, f)   # UpdateKinematics / SetContact / SetTaskSpace
    wbc.solve()                                        # CalcContactConstraint .. CalcContactRedistribute, one launch
    tau = wbc.get("tau_total")

All arithmetic happens in libdwbc_hip.so (hand-written HIP, gfx950).  torch is only used, optionally, to own
device buffers and streams (bind_tensor) and for torch.distributed in bench.py.
"""
import ctypes as C

import numpy as np

from . import _lib

CONTACT_6D = 0
TASK_LINK_6D, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME = 0, 1, 2
TASK_LINK_POSITION, TASK_LINK_POSITION_COM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME = 3, 4, 5
TASK_LINK_ROTATION, TASK_LINK_ROTATION_CUSTOM_FRAME = 6, 7
SOLVE_HQP, SOLVE_INIT, SOLVE_REDUCED = 1, 2, 4

# field ids of include/dwbc_batch.h
FIELDS = dict(
    in_q=0, in_contact=1, in_fstar=2, tau=10, wrench=11, status=12, diag=13,
    tau_grav=20, tau_task=21, tau_contact=22, tau_total=23,
    A=30, A_inv=31, J_C=32, Lambda_c=33, J_C_INV_T=34, A_inv_N_C=35, W_inv=36, NwJw=37, G=38, P_C=39,
    link_R=40, link_p=41, fstar_qp=42, contact_qp=43, cf_redis=44, J_task=45, Lambda_task=46, J_kt=47, qp_viol=48,
    CMM=50, com=51, com_inertia=52, J_com=53, B=54, link_v=55, link_w=56, contact_pos=57, contact_rot=58, zmp=59,
    A_R=60, A_R_inv=61, G_R=62, J_I_nc=63, J_I_nc_inv_T=64,
)


class DwbcError(RuntimeError):
    pass


def _check(ok):
    if not ok:
        raise DwbcError(_lib.last_error())


def tree_tag(parents):
    """name tag of a tree-specific kernel pack: FNV-1a over the parent table as little-endian 32-bit words (dwbc_capi.hip: tree_tag)"""
    h = 2166136261
    for p in parents:
        for b in int(max(p, 0)).to_bytes(4, "little"):
            h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    return f"{h:08x}"


def build_pack(model_or_ndof, nb=None, quiet=True, tree=False):
    """Compile the cycle kernels for a model size other than TOCABI's (libdwbc_amd/csrc/dwbc_pack.hip -> libdwbc_pack_<N>_<NB>.so next
    to libdwbc_hip.so; about two minutes with hipcc, nothing to do when it is up to date).  dwbc_batch_create loads it by itself.
    tree=True (needs a Model): a pack for exactly this model's kinematic tree (libdwbc_pack_<N>_<NB>_t<tag>.so) -- the tree-sparse
    A^-1 sweep and compile-time round counts the built-in TOCABI kernels have; it is preferred over the generic pack of the size."""
    import os
    import subprocess

    if nb is None:
        n, nb = int(model_or_ndof.ndof), int(model_or_ndof.nb)
    else:
        n, nb = int(model_or_ndof), int(nb)
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    cmd = ["make", "-C", csrc, "pack", f"N={n}", f"NB={nb}"]
    name = f"libdwbc_pack_{n}_{nb}.so"
    if tree:
        parents = [max(int(p), 0) for p in model_or_ndof.arrays()["parent"]]
        tag = tree_tag(parents)
        cmd += ["PARENTS=" + ",".join(str(p) for p in parents), f"TAG={tag}"]
        name = f"libdwbc_pack_{n}_{nb}_t{tag}.so"
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL if quiet else None)
    return os.path.join(os.path.dirname(csrc), name)


class Model:
    def __init__(self, handle):
        self._L = _lib.load()
        self._h = handle
        self.nb = self._L.dwbc_model_num_links(handle)
        self.ndof = self._L.dwbc_model_system_dof(handle)
        self.total_mass = self._L.dwbc_model_total_mass(handle)

    @classmethod
    def from_urdf(cls, path, floating=True):
        L = _lib.load()
        h = L.dwbc_model_create_from_urdf(str(path).encode(), 1 if floating else 0)
        if not h:
            raise DwbcError(_lib.last_error())
        return cls(h)

    @classmethod
    def from_arrays(cls, m):
        L = _lib.load()
        arrs = [np.ascontiguousarray(m["parent"], np.int32)] + [
            np.ascontiguousarray(m[k], np.float64) for k in ("R_T", "p_T", "axis", "mass", "com", "inertia")
        ]
        h = L.dwbc_model_create_from_arrays(int(m["nb"]), *[a.ctypes.data for a in arrs])
        if not h:
            raise DwbcError(_lib.last_error())
        return cls(h)

    def link_id(self, name):
        return self._L.dwbc_model_link_id(self._h, name.encode())

    # ---- init-time model surgery (RobotData::DeleteLink / AddLink / ChangeLinkToFixedJoint / ChangeLinkInertia, src/dwbc.cpp:1764-2382,
    #      2707-2730): each returns a NEW Model; a model of another size runs on its own kernel pack (build_pack)
    def _link(self, link):
        i = self.link_id(link) if isinstance(link, str) else int(link)
        if i < 0:
            raise DwbcError(f"no link named {link}")
        return i

    def _new(self, h):
        if not h:
            raise DwbcError(_lib.last_error())
        return Model(h)

    def delete_link(self, link):
        return self._new(self._L.dwbc_model_delete_link(self._h, self._link(link)))

    def add_link(self, parent, name, joint_type, joint_axis, joint_rotm, joint_trans, mass, com, inertia):
        """joint_type: 0 fixed (joined to the parent link), 1 revolute (a new last link; the parent must be the last link or one of
        its ancestors).  joint_rotm: rotation child -> parent of the joint frame"""
        a = [np.ascontiguousarray(x, np.float64) for x in (joint_axis, joint_rotm, joint_trans, com, inertia)]
        return self._new(self._L.dwbc_model_add_link(self._h, self._link(parent), str(name).encode(), int(joint_type), a[0].ctypes.data, a[1].ctypes.data,
                                                     a[2].ctypes.data, float(mass), a[3].ctypes.data, a[4].ctypes.data))

    def change_link_to_fixed_joint(self, link):
        return self._new(self._L.dwbc_model_change_link_to_fixed_joint(self._h, self._link(link)))

    def change_link_inertia(self, link, com_inertia, com_position, com_mass):
        a = [np.ascontiguousarray(x, np.float64) for x in (com_inertia, com_position)]
        return self._new(self._L.dwbc_model_change_link_inertia(self._h, self._link(link), a[0].ctypes.data, a[1].ctypes.data, float(com_mass)))

    def link_name(self, i):
        return self._L.dwbc_model_link_name(self._h, i).decode()

    def arrays(self):
        nb = self.nb
        out = dict(parent=np.zeros(nb, np.int32), R_T=np.zeros((nb, 3, 3)), p_T=np.zeros((nb, 3)), axis=np.zeros((nb, 3)),
                   mass=np.zeros(nb), com=np.zeros((nb, 3)), inertia=np.zeros((nb, 3, 3)))
        self._L.dwbc_model_get_arrays(self._h, *[out[k].ctypes.data for k in ("parent", "R_T", "p_T", "axis", "mass", "com", "inertia")])
        out["nb"], out["ndof"] = nb, self.ndof
        return out

    def __del__(self):
        try:
            if self._h:
                self._L.dwbc_model_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Batch:
    def __init__(self, model, B, device=0, dtype="f64"):
        self._L = _lib.load()
        self.model = model
        self.B = int(B)
        self.n = model.ndof
        self.m = model.ndof - 6
        self._h = self._L.dwbc_batch_create(model._h, self.B, int(device), {"f64": 0, "f32": 1}[dtype])
        if not self._h:
            raise DwbcError(_lib.last_error())
        self.n_contacts = 0
        self._keep = {}

    # ---- setup (shared by all instances)
    def add_contact(self, link, point, lx, ly, mu=0.2, mu_z=0.2, contact_type=CONTACT_6D):
        p = np.ascontiguousarray(point, np.float64)
        i = self._L.dwbc_batch_add_contact(self._h, int(link), int(contact_type), p.ctypes.data, lx, ly, mu, mu_z)
        if i < 0:
            raise DwbcError(_lib.last_error())
        self.n_contacts = i + 1
        return i

    def add_task(self, level, mode, link, point=(0.0, 0.0, 0.0)):
        p = np.ascontiguousarray(point, np.float64)
        _check(self._L.dwbc_batch_add_task(self._h, int(level), int(mode), int(link), p.ctypes.data))

    def set_torque_limit(self, lim):
        if lim is None:
            _check(self._L.dwbc_batch_set_torque_limit(self._h, None))
        else:
            t = np.ascontiguousarray(lim, np.float64)
            assert t.shape == (self.m,)
            _check(self._L.dwbc_batch_set_torque_limit(self._h, t.ctypes.data))

    @property
    def fstar_size(self):
        return self._L.dwbc_batch_fstar_size(self._h)

    def task_dof(self, level):
        return self._L.dwbc_batch_task_dof(self._h, level)

    # ---- per-cycle inputs (host arrays)
    def set_state(self, q, qdot=None, qddot=None):
        q = np.ascontiguousarray(q, np.float64)
        assert q.shape == (self.B, self.n + 1), q.shape
        qd = None
        if qdot is not None:
            qd = np.ascontiguousarray(qdot, np.float64)
            assert qd.shape == (self.B, self.n), qd.shape
        _check(self._L.dwbc_batch_set_state(self._h, q.ctypes.data, qd.ctypes.data if qd is not None else None, None))

    def add_custom_task(self, level, task_dof):
        """AddTaskSpace(heirarchy, TASK_CUSTOM, task_dof) (reference include/dwbc.h:318)"""
        _check(self._L.dwbc_batch_add_custom_task(self._h, int(level), int(task_dof)))

    def set_custom_task(self, level, fstar, J):
        """SetTaskSpace(heirarchy, f*, J_task) (reference include/dwbc.h:333): fstar (B, t), J (B, t, n)"""
        J = np.ascontiguousarray(J, np.float64)
        t = self.task_dof(level)
        assert J.shape == (self.B, t, self.n), J.shape
        f = None
        if fstar is not None:
            f = np.ascontiguousarray(fstar, np.float64)
            assert f.shape == (self.B, t), f.shape
        _check(self._L.dwbc_batch_set_custom_task(self._h, int(level), f.ctypes.data if f is not None else None, J.ctypes.data))

    def set_task_gain(self, level, link_index, pos_p, pos_d, pos_a, rot_p, rot_d, rot_a=(0, 0, 0)):
        """TaskLink::SetTaskGain (reference include/dwbc_task.h:108)"""
        arrs = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), (3,))) for a in (pos_p, pos_d, pos_a, rot_p, rot_d, rot_a)]
        _check(self._L.dwbc_batch_set_task_gain(self._h, int(level), int(link_index), *[a.ctypes.data for a in arrs]))

    def set_trajectory(self, level, link_index, traj):
        """per-instance trajectory records (B, 34): SetTrajectoryQuintic + SetTrajectoryRotation; None clears"""
        if traj is None:
            _check(self._L.dwbc_batch_set_trajectory(self._h, int(level), int(link_index), None))
            return
        t = np.ascontiguousarray(traj, np.float64)
        assert t.shape == (self.B, 34), t.shape
        _check(self._L.dwbc_batch_set_trajectory(self._h, int(level), int(link_index), t.ctypes.data))

    def set_control_time(self, t):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, np.float64), (self.B,)))
        _check(self._L.dwbc_batch_set_control_time(self._h, t.ctypes.data))

    def copy_kinematics_to(self, target):
        """RobotData::CopyKinematicsData(target) (reference include/dwbc.h:375)"""
        _check(self._L.dwbc_batch_copy_kinematics(target._h, self._h))
        target.n_contacts = self.n_contacts

    def set_max_active_contacts(self, n):
        """Contacts that may be active at once in one instance: 2 (default, the product kernels) or 3 -- every solve of the batch then
        runs the general-contact kernel (feet + one hand, ...; hqp = true, link tasks) and ``get("wrench")`` is (B, 6 n).  The
        reference stacks any number of flagged contacts (src/dwbc.cpp:445-453)."""
        _check(self._L.dwbc_batch_set_max_active_contacts(self._h, int(n)))

    @property
    def max_active_contacts(self):
        return int(self._L.dwbc_batch_max_active_contacts(self._h))

    def set_contact(self, flags):
        f = np.ascontiguousarray(flags, np.uint8)
        assert f.shape == (self.B, self.n_contacts), f.shape
        _check(self._L.dwbc_batch_set_contact(self._h, f.ctypes.data))

    def set_fstar(self, level, fstar):
        f = np.ascontiguousarray(fstar, np.float64)
        assert f.shape == (self.B, self.task_dof(level)), f.shape
        _check(self._L.dwbc_batch_set_fstar(self._h, int(level), f.ctypes.data))

    def set_fstar_all(self, fstar):
        if fstar.flags["C_CONTIGUOUS"] and fstar.dtype == np.float64 and fstar.ctypes.data == (self._L.dwbc_batch_host_ptr(self._h, FIELDS["in_fstar"]) or 0):
            # the mirror itself (host_view("in_fstar")), filled in place: mark every level as new without copying
            off = 0
            for lv in range(64):
                if off >= fstar.shape[1]:
                    break
                _check(self._L.dwbc_batch_set_fstar(self._h, lv, C.c_void_p(fstar.ctypes.data + 8 * off)))
                off += self.task_dof(lv)
            return
        off = 0
        lv = 0
        while off < fstar.shape[1]:
            t = self.task_dof(lv)
            self.set_fstar(lv, fstar[:, off : off + t])
            off += t
            lv += 1

    def host_view(self, field):
        """numpy view of the page-locked host mirror of an input field ("in_q", "in_contact", "in_fstar"): fill it in place and pass
        it to set_state / set_contact / set_fstar_all -- the host-side copy is skipped, the upload is one asynchronous transfer"""
        shape = (self.B,) + tuple(self._SHAPES[field](self))
        dt = np.uint8 if field == "in_contact" else np.float64
        p = self._L.dwbc_batch_host_ptr(self._h, FIELDS[field])
        if not p:
            raise DwbcError(f"{field} has no host mirror yet (add the contacts / tasks first)")
        n = int(np.prod(shape))
        buf = (C.c_uint8 * n if dt == np.uint8 else C.c_double * n).from_address(p)
        return np.frombuffer(buf, dtype=dt).reshape(shape)

    # ---- zero-copy device plumbing (torch owns the memory / stream)
    def bind_tensor(self, field, tensor):
        assert tensor.is_cuda and tensor.is_contiguous()
        nbytes = self._L.dwbc_batch_field_bytes(self._h, FIELDS[field])
        assert tensor.numel() * tensor.element_size() == nbytes, (field, tensor.shape, nbytes)
        self._keep[field] = tensor
        _check(self._L.dwbc_batch_bind_device(self._h, FIELDS[field], C.c_void_p(tensor.data_ptr())))

    def set_stream(self, stream_handle):
        _check(self._L.dwbc_batch_set_stream(self._h, C.c_void_p(stream_handle)))

    def enable_dump(self, on=True):
        _check(self._L.dwbc_batch_enable_dump(self._h, 1 if on else 0))

    # ---- the cycle
    def solve(self, hqp=True, init=True, reduced=False):
        """reduced=True: the Reduced* call sequence of the reference (include/dwbc.h:411-416) instead of the full model"""
        _check(self._L.dwbc_batch_solve(self._h, (SOLVE_HQP if hqp else 0) | (SOLVE_INIT if init else 0) | (SOLVE_REDUCED if reduced else 0)))

    def sync(self):
        _check(self._L.dwbc_batch_sync(self._h))

    def time_solves(self, steps, reduced=False):
        ms = C.c_float(0)
        _check(self._L.dwbc_batch_time_solves(self._h, SOLVE_HQP | SOLVE_INIT | (SOLVE_REDUCED if reduced else 0), int(steps), C.byref(ms)))
        return ms.value

    def launch_info(self):
        t, l = C.c_int(0), C.c_int(0)
        self._L.dwbc_batch_launch_info(self._h, C.byref(t), C.byref(l))
        return t.value, l.value

    def kernel_name(self):
        return self._L.dwbc_batch_kernel_name(self._h).decode()

    _SHAPES = dict(
        tau=lambda s: (3, s.m), wrench=lambda s: (6 * s.max_active_contacts,), status=lambda s: (), diag=lambda s: (90,),
        tau_grav=lambda s: (s.m,), tau_task=lambda s: (s.m,), tau_contact=lambda s: (s.m,), tau_total=lambda s: (s.m,),
        A=lambda s: (s.n, s.n), A_inv=lambda s: (s.n, s.n), A_inv_N_C=lambda s: (s.n, s.n), J_C=lambda s: (12, s.n),
        J_C_INV_T=lambda s: (12, s.n), Lambda_c=lambda s: (144,), W_inv=lambda s: (s.m, s.m), NwJw=lambda s: (s.m, 6),
        G=lambda s: (s.n,), P_C=lambda s: (12,), link_R=lambda s: (48, 3, 3), link_p=lambda s: (48, 3),
        fstar_qp=lambda s: (4, 6), contact_qp=lambda s: (4, 6), cf_redis=lambda s: (6,), J_task=lambda s: (4, 6 * s.n),
        Lambda_task=lambda s: (4, 36), J_kt=lambda s: (4, s.m * 6), qp_viol=lambda s: (5,),
        CMM=lambda s: (6, s.n), com=lambda s: (3,), com_inertia=lambda s: (3, 3), J_com=lambda s: (6, s.n),
        B=lambda s: (s.n,), link_v=lambda s: (48, 3), link_w=lambda s: (48, 3),
        contact_pos=lambda s: (2, 3), contact_rot=lambda s: (2, 3, 3), zmp=lambda s: (3, 3),
        A_R=lambda s: (24, 24), A_R_inv=lambda s: (24, 24), G_R=lambda s: (24,), J_I_nc=lambda s: (6, s.n - 12), J_I_nc_inv_T=lambda s: (6, s.n - 12),
        in_q=lambda s: (s.n + 1,), in_contact=lambda s: (s.n_contacts,), in_fstar=lambda s: (s.fstar_size,),
    )

    def get(self, field):
        fid = FIELDS[field]
        shape = (self.B,) + tuple(self._SHAPES[field](self))
        dt = np.int32 if field in ("status", "diag") else (np.uint8 if field == "in_contact" else np.float64)
        out = np.zeros(shape, dtype=dt)
        nbytes = self._L.dwbc_batch_field_bytes(self._h, fid)
        assert nbytes == out.nbytes, (field, nbytes, out.nbytes)
        _check(self._L.dwbc_batch_get(self._h, fid, out.ctypes.data, out.nbytes))
        return out

    def close(self):
        if self._h:
            self._L.dwbc_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
