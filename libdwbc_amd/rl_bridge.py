"""Vectorised counterpart of the reference's RL bridge (src/pybind/rl_wbc_bridge.cpp:7-139): one RlWBCBridge object
drives `n_envs` TOCABI whole-body controllers in a single kernel launch per CalcTorque().

Same call names, argument meaning and output type as the reference class, with a leading environment axis:

    reference (one env)                                this class (n_envs)
    RlWBCBridge(env_id)                                RlWBCBridge(n_envs, urdf=..., device=0)
    UpdateKinematics(qpos[40], qvel[39], qacc[39])     UpdateKinematics(qpos[B,40], qvel[B,39], qacc[B,39])
    SetContact(left, right)                            SetContact(left[B] | bool, right[B] | bool)
    SetTaskSpace(h, f_star[t])                         SetTaskSpace(h, f_star[B,t] | f_star[t])
    CalcTorque(); getTorqueCommand() -> float[33]      CalcTorque(); getTorqueCommand() -> float32[B,33]

qpos is MuJoCo order (pos3, quaternion w x y z, 33 joints); the reorder to RBDL order (pos3, quat xyz, joints, quat w
last) follows rl_wbc_bridge.cpp:88-98.  The controller setup (feet contacts, pelvis 6D + upper-body rotation tasks,
torque limit 300) is rl_wbc_bridge.cpp:19-43.  numpy arrays or torch CUDA tensors are accepted; with CUDA tensors the
reorder runs on the device and nothing crosses PCIe.

`hqp`: the reference's CalcTorque() (rl_wbc_bridge.cpp:123-129) sets `task_init = false` and calls
`CalcTaskControlTorque(task_init)` / `CalcContactRedistribute(task_init)`, whose FIRST parameter is `hqp`
(include/dwbc.h:349,298) -- so the reference bridge effectively runs hqp = false (plain task hierarchy + closed-form
two-foot redistribution, no QP, no torque-limit or cone rows) with init = true.  hqp=False (the default) reproduces
that; hqp=True runs the QP cascade every step (what tests/dwbc_test.cpp exercises and what the goldens pin).
"""
import numpy as np

from .batch import CONTACT_6D, TASK_LINK_6D, TASK_LINK_ROTATION, Batch, Model

LEFT_FOOT, RIGHT_FOOT, PELVIS, UPPER_BODY = 6, 12, 0, 15
FOOT_POINT = (0.03, 0.0, -0.1585)


def mujoco_to_rbdl_q(qpos):
    """(…,40) MuJoCo qpos -> (…,40) RBDL q (reference src/pybind/rl_wbc_bridge.cpp:88-98).  numpy or torch."""
    if isinstance(qpos, np.ndarray):
        q = np.empty_like(qpos, dtype=np.float64)
    else:
        import torch

        q = torch.empty_like(qpos, dtype=torch.float64)
    q[..., 0:3] = qpos[..., 0:3]
    q[..., 3:6] = qpos[..., 4:7]
    q[..., 6:39] = qpos[..., 7:40]
    q[..., 39] = qpos[..., 3]
    return q


class RlWBCBridge:
    def __init__(self, n_envs, urdf, device=0, torque_limit=300.0, dtype="f64", hqp=False):
        """dtype="f32": the fp32 kernels -- the bridge returns float32 torques either way.
        hqp=False: the reference bridge's effective behaviour (module docstring); hqp=True: QP cascade every step.
        torque_limit only enters the QP rows: with hqp=False (the default, as in the reference bridge) it has NO effect -- the
        plain hierarchy and the closed-form redistribution know no torque limit or friction cone.  With hqp=True every step after
        the first is a hot start (init=False), which runs the full kernel build (working sets carried in HBM), about 10 % slower
        per launch than the lean build a cold solve uses."""
        self.n_envs = int(n_envs)
        self.hqp = bool(hqp)
        self.model = Model.from_urdf(urdf)
        self.wbc = Batch(self.model, self.n_envs, device=device, dtype=dtype)
        self.model_dof = self.wbc.m
        self.wbc.add_contact(LEFT_FOOT, FOOT_POINT, 0.15, 0.075, contact_type=CONTACT_6D)
        self.wbc.add_contact(RIGHT_FOOT, FOOT_POINT, 0.15, 0.075, contact_type=CONTACT_6D)
        self.wbc.add_task(0, TASK_LINK_6D, PELVIS)
        self.wbc.add_task(1, TASK_LINK_ROTATION, UPPER_BODY)
        self.wbc.set_torque_limit(np.full(self.model_dof, float(torque_limit)))
        self._device = device
        self._q_dev = None
        self._fstar = np.zeros((self.n_envs, self.wbc.fstar_size))
        self._fstar_dev = None
        self.task_init = True

    # reference rl_wbc_bridge.cpp:82-107.  qvel / qacc are accepted for signature parity; the OSF torque path does not
    # depend on them (they only enter RobotData::B_, which CalcTaskControlTorque never reads).
    def UpdateKinematics(self, qpos, qvel=None, qacc=None):
        if isinstance(qpos, np.ndarray) or not getattr(qpos, "is_cuda", False):
            q = mujoco_to_rbdl_q(np.asarray(qpos, dtype=np.float64).reshape(self.n_envs, 40))
            self.wbc.set_state(q)
        else:
            import torch

            if self._q_dev is None:
                self._q_dev = torch.empty((self.n_envs, 40), dtype=torch.float64, device=qpos.device)
                self.wbc.bind_tensor("in_q", self._q_dev)
            self._q_dev.copy_(mujoco_to_rbdl_q(qpos.reshape(self.n_envs, 40)))

    # reference rl_wbc_bridge.cpp:109-114 (CalcContactConstraint / CalcTaskSpace run inside the fused launch)
    def SetContact(self, left, right):
        f = np.empty((self.n_envs, 2), dtype=np.uint8)
        f[:, 0] = np.asarray(left, dtype=bool)
        f[:, 1] = np.asarray(right, dtype=bool)
        self.wbc.set_contact(f)

    # reference rl_wbc_bridge.cpp:116-121
    def SetTaskSpace(self, heirarchy, f_star):
        t = self.wbc.task_dof(heirarchy)
        if not isinstance(f_star, np.ndarray) and getattr(f_star, "is_cuda", False):
            f_star = f_star.detach().cpu().numpy()
        f = np.broadcast_to(np.asarray(f_star, dtype=np.float64).reshape(-1, t), (self.n_envs, t))
        self.wbc.set_fstar(heirarchy, np.ascontiguousarray(f))

    # reference rl_wbc_bridge.cpp:123-129
    def CalcTorque(self):
        # hqp = false: the closed-form branch has no QP state, `init` is irrelevant there (the reference passes true)
        self.wbc.solve(hqp=self.hqp, init=True if not self.hqp else self.task_init)
        self.task_init = False

    # reference rl_wbc_bridge.cpp:131-139: float vector of torque_grav_ + torque_task_ + torque_contact_
    def getTorqueCommand(self):
        return self.wbc.get("tau_total").astype(np.float32)

    def status(self):
        """Per-env return flags of the cycle (bit meanings: include/dwbc_batch.h)."""
        return self.wbc.get("status")

    def Reset(self):
        self.task_init = True

    def reflectAction(self, action):  # empty in the reference as well (rl_wbc_bridge.cpp:54-80)
        return None
