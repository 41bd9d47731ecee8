/*
 * dwbc_batch.h -- C-ABI of the MI355X-native batched libdwbc hot path (libdwbc_hip.so).
 *
 * The reference (saga0619/libdwbc) has no FFI: its boundary is the C++ class DWBC::RobotData
 * (reference include/dwbc.h:59-430).  Each entry point below replaces the RobotData method cited next to
 * it, for B independent robot instances at once; include/dwbc_amd.hpp is the header-only C++ facade that
 * gives the RobotData names/arguments back on top of this ABI (B = 1 reproduces single-robot behaviour).
 *
 * Conventions (same as the reference): q = [x y z | qx qy qz | joints | qw] (size ndof+1), qdot/qddot size
 * ndof with the base angular velocity in the body frame; Jacobian rows / wrenches are [linear; angular] in
 * the world frame; int returns are 1 = ok, 0 = failure (no exceptions); failures leave a message in
 * dwbc_last_error().  Host arrays are instance-major (AoS), double precision.
 * Not thread-safe per object (one dwbc_batch per thread), like RobotData.
 */
#ifndef DWBC_BATCH_H
#define DWBC_BATCH_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dwbc_model dwbc_model;
typedef struct dwbc_batch dwbc_batch;

/* contact types: reference include/dwbc_contact_constraint.h:19-25 */
enum { DWBC_CONTACT_6D = 0, DWBC_CONTACT_POINT = 1, DWBC_CONTACT_LINK = 2, DWBC_CONTACT_LINE = 3 };
/* task link modes: reference include/dwbc_task.h:23-33 */
enum {
    DWBC_TASK_LINK_6D = 0, DWBC_TASK_LINK_6D_COM_FRAME, DWBC_TASK_LINK_6D_CUSTOM_FRAME,
    DWBC_TASK_LINK_POSITION, DWBC_TASK_LINK_POSITION_COM_FRAME, DWBC_TASK_LINK_POSITION_CUSTOM_FRAME,
    DWBC_TASK_LINK_ROTATION, DWBC_TASK_LINK_ROTATION_CUSTOM_FRAME
};
/* arithmetic type of the kernels.  DWBC_F32 runs the same kernel source in single precision; every buffer at this boundary
 * (host arrays, bound device buffers) stays double and is converted inside the kernel.  Accuracy envelope: DESIGN.md §8 */
enum { DWBC_F64 = 0, DWBC_F32 = 1 };
/* DWBC_SOLVE_HQP: CalcTaskControlTorque(hqp) / CalcContactRedistribute(hqp) with hqp = true (the QP cascade); without the bit the
 * plain hierarchy and the closed-form two-contact redistribution run (src/dwbc.cpp:856-873, 1570-1619), full model only.
 * DWBC_SOLVE_REDUCED: the Reduced* call sequence (ReducedDynamicsCalculate, ReducedCalcContactConstraint,
 * ReducedCalcGravCompensation, ReducedCalcTaskSpace, ReducedCalcTaskControlTorque, ReducedCalcContactRedistribute --
 * reference include/dwbc.h:411-416, tests/sp_test/redu_dyn_test.cpp:263-298) instead of the full-model sequence.
 * DWBC_SOLVE_INIT: the `init` argument of CalcTaskControlTorque / CalcContactRedistribute.  Set: cold start (qpOASES init).
 * Clear: hot start (qpOASES hotstart, src/qp_wrapper.cpp:249-296) -- every QP of the full-model path first visits the rows of
 * its working set from the previous dwbc_batch_solve on this batch (kept in HBM); the returned point is the same canonical
 * point, only the number of active-set steps changes.  The first solve of a batch always runs cold.  The failure fallback of
 * SolveQPoases (fresh problem, setToReliable, nWSR x 10, src/qp_wrapper.cpp:298-339) has no counterpart to switch to: the
 * device solver has no heuristic options, its step budget is the reference's nWSR + 10 nWSR. */
enum { DWBC_SOLVE_HQP = 1, DWBC_SOLVE_INIT = 2, DWBC_SOLVE_REDUCED = 4 };

/* fields for dwbc_batch_get / dwbc_batch_bind_device.  Shapes are per instance, row-major. */
enum dwbc_field {
    /* inputs (bindable) */
    DWBC_IN_Q = 0,        /* (ndof+1)        f64  -- UpdateKinematics(q, ...)          */
    DWBC_IN_CONTACT = 1,  /* (n_contacts)    u8   -- SetContact(...)                   */
    DWBC_IN_FSTAR = 2,    /* (sum task dof)  f64  -- SetTaskSpace(level, f*) concatenated */
    /* outputs (bindable) */
    DWBC_TAU = 10,        /* (3, m) f64: torque_grav_, torque_task_, torque_contact_  (include/dwbc.h:115-117) */
    DWBC_WRENCH = 11,     /* (12; 18 after dwbc_batch_set_max_active_contacts(b, 3)) f64: getContactForce(tau_total), zero padded (src/dwbc.cpp:891-896) */
    DWBC_STATUS = 12,     /* (1)    i32: 1 ok / 0 failed                               */
    DWBC_DIAG = 13,       /* (90)   i32: stage status, QP iterations, working sets, stage stamps */
    /* derived getters (host only) */
    DWBC_TAU_GRAV = 20, DWBC_TAU_TASK = 21, DWBC_TAU_CONTACT = 22, DWBC_TAU_TOTAL = 23,
    /* intermediates, available after a solve with dwbc_batch_enable_dump(b, 1) -- the RobotData public fields */
    DWBC_A = 30,          /* (n, n)   A_        */
    DWBC_A_INV = 31,      /* (n, n)   A_inv_    */
    DWBC_J_C = 32,        /* (12, n)  J_C       */
    DWBC_LAMBDA_C = 33,   /* (12, 12) Lambda_contact (row stride = active contact dof) */
    DWBC_J_C_INV_T = 34,  /* (12, n)  J_C_INV_T */
    DWBC_A_INV_N_C = 35,  /* (n, n)   A_inv_N_C */
    DWBC_W_INV = 36,      /* (m, m)   W_inv     */
    DWBC_NWJW = 37,       /* (m, 6)   NwJw      */
    DWBC_G = 38,          /* (n)      G_        */
    DWBC_P_C = 39,        /* (12)     P_C       */
    DWBC_LINK_R = 40,     /* (48, 9)  link_[i].rotm */
    DWBC_LINK_P = 41,     /* (48, 3)  link_[i].xpos */
    DWBC_FSTAR_QP = 42,   /* (4, 6)   ts_[l].f_star_qp_  */
    DWBC_CONTACT_QP = 43, /* (4, 6)   ts_[l].contact_qp_ */
    DWBC_CF_REDIS = 44,   /* (6)      cf_redis_qp_       */
    DWBC_J_TASK = 45,     /* (4, 6, n) ts_[l].J_task_ (row stride n)       */
    DWBC_LAMBDA_TASK = 46,/* (4, 36)  ts_[l].Lambda_task_ (row stride t_l) */
    DWBC_J_KT = 47,       /* (4, m*6) ts_[l].J_kt_ (row stride t_l)        */
    DWBC_QP_VIOL = 48,    /* (5)      worst normalised slack of each QP's returned point */
    DWBC_DUMP_RAW = 49,   /* whole dump record */
    DWBC_CMM = 50,        /* (6, n)   CMM_                      (src/dwbc.cpp:336) */
    DWBC_COM = 51,        /* (3)      com_pos                   (src/dwbc.cpp:322) */
    DWBC_COM_INERTIA = 52,/* (3, 3)   link_.back().inertia      (src/dwbc.cpp:343) */
    DWBC_J_COM = 53,      /* (6, n)   link_.back().jac_com_     (src/dwbc.cpp:352) */
    /* velocity-dependent outputs of UpdateKinematics: need a qdot in dwbc_batch_set_state */
    DWBC_B = 54,          /* (n)      B_ = C qdot + g (RNEA)    (src/dwbc.cpp:343-344) */
    DWBC_LINK_V = 55,     /* (48, 3)  link_[i].v                (src/link.cpp:87) */
    DWBC_LINK_W = 56,     /* (48, 3)  link_[i].w                (src/link.cpp:88) */
    DWBC_CONTACT_POS = 57,/* (2, 3)   cc_[i].xc_pos of the active contacts (src/contact_constraint.cpp:53) */
    DWBC_CONTACT_ROT = 58,/* (2, 9)   cc_[i].rotm */
    DWBC_ZMP = 59,        /* (3, 3)   getZMP(getContactForce(tau_total)) then cc_[i].zmp_pos (src/dwbc.cpp:898-939) */
    /* the reduced (centroidal) model, after a solve with DWBC_SOLVE_REDUCED (src/dwbc.cpp:2932-2988); RS = vc_dof + 6 <= 24 */
    DWBC_A_R = 60,        /* (24, 24) A_R, row stride 24, zero beyond RS */
    DWBC_A_R_INV = 61,    /* (24, 24) A_R_inv */
    DWBC_G_R = 62,        /* (24)     G_R */
    DWBC_J_I_NC = 63,     /* (6, n-12) J_I_nc_, row stride n - 12, zero beyond nc_dof */
    DWBC_J_I_NC_INV_T = 64 /* (6, n-12) J_I_nc_inv_T */
};

const char *dwbc_last_error(void);
int dwbc_device_count(void);

/* ---- model: RobotData::LoadModelData(urdf, floating, verbose)  reference include/dwbc.h:237, src/dwbc.cpp:102-252 */
dwbc_model *dwbc_model_create_from_urdf(const char *urdf_path, int floating_base);
/* same tree given as arrays (R_T: nb x 9 child->parent joint-frame rotation, inertia: nb x 9 about the com) */
dwbc_model *dwbc_model_create_from_arrays(int nb, const int32_t *parent, const double *R_T, const double *p_T,
                                          const double *axis, const double *mass, const double *com,
                                          const double *inertia);
void dwbc_model_destroy(dwbc_model *m);
int dwbc_model_num_links(const dwbc_model *m);   /* link_num_   */
int dwbc_model_system_dof(const dwbc_model *m);  /* system_dof_ */
double dwbc_model_total_mass(const dwbc_model *m); /* total_mass_ */
int dwbc_model_link_id(const dwbc_model *m, const char *name); /* case-insensitive, -1 if absent (src/dwbc.cpp:397-406) */
const char *dwbc_model_link_name(const dwbc_model *m, int link);
int dwbc_model_get_arrays(const dwbc_model *m, int32_t *parent, double *R_T, double *p_T, double *axis, double *mass,
                          double *com, double *inertia);
/* ---- init-time model surgery (reference include/dwbc.h:206-226, src/dwbc.cpp:1764-2382, 2707-2730).  Each call returns a NEW
 * model (NULL + dwbc_last_error on refusal) and leaves its argument alone: destroy both when done.  A model of another size
 * runs on its own kernel pack (`make -C libdwbc_amd/csrc pack N=.. NB=..`).
 *   DeleteLink(link)                  the link and all its descendants; the links after them move down
 *   AddLink(parent, name, joint_type, axis, joint_rotm, joint_trans, mass, com, inertia)
 *                                     joint_type 0 = JOINT_FIXED: the body is joined to the parent link (RBDL Body::Join), no new link;
 *                                     1 = JOINT_REVOLUTE: a new LAST link whose joint is the last coordinate -- accepted when the parent
 *                                     is the last link or one of its ancestors (the kernels need the depth-first numbering kept)
 *   ChangeLinkToFixedJoint(link)      = DeleteLink(link) + AddLink(fixed) of the link's own body, as the reference does it
 *                                     (src/dwbc.cpp:2360-2382: the link's descendants are deleted with it)
 *   ChangeLinkInertia(link, com_inertia, com_position, com_mass)
 * joint_rotm: 3x3 row-major rotation child -> parent of the joint frame, joint_trans its origin in the parent frame */
dwbc_model *dwbc_model_delete_link(const dwbc_model *m, int link);
dwbc_model *dwbc_model_add_link(const dwbc_model *m, int parent_link, const char *link_name, int joint_type, const double *joint_axis,
                                const double *joint_rotm, const double *joint_trans, double body_mass, const double *com_position,
                                const double *inertia);
dwbc_model *dwbc_model_change_link_to_fixed_joint(const dwbc_model *m, int link);
dwbc_model *dwbc_model_change_link_inertia(const dwbc_model *m, int link, const double *com_inertia, const double *com_position, double com_mass);

/* ---- batch of B RobotData instances sharing one model and one contact/task setup ---- */
dwbc_batch *dwbc_batch_create(const dwbc_model *m, int B, int device, int dtype);
void dwbc_batch_destroy(dwbc_batch *b);
int dwbc_batch_size(const dwbc_batch *b);

/* AddContactConstraint(link, type, point, normal, lx, ly)  include/dwbc.h:259 ; SetFrictionRatio src/contact_constraint.cpp:93.
 * returns the contact index or -1 */
int dwbc_batch_add_contact(dwbc_batch *b, int link, int contact_type, const double point[3], double lx, double ly,
                           double mu, double mu_z);
int dwbc_batch_clear_contacts(dwbc_batch *b);                               /* ClearContactConstraint */
/* AddTaskSpace(level, mode, link, point) include/dwbc.h:319 (same level twice appends a link, src/dwbc.cpp:592-600) */
int dwbc_batch_add_task(dwbc_batch *b, int level, int mode, int link, const double point[3]);
/* AddTaskSpace(heirarchy, TASK_CUSTOM, task_dof) include/dwbc.h:318 and SetTaskSpace(heirarchy, f*, J_task) include/dwbc.h:333:
 * a level whose Jacobian the caller supplies, per instance (fstar: B x task_dof or NULL to keep, J: B x task_dof x ndof row-major) */
int dwbc_batch_add_custom_task(dwbc_batch *b, int level, int task_dof);
int dwbc_batch_set_custom_task(dwbc_batch *b, int level, const double *fstar, const double *J);
int dwbc_batch_clear_tasks(dwbc_batch *b);                                  /* ClearTaskSpace */
/* ---- on-device task reference (reference src/task.cpp:223-339, src/dwbc.cpp:708-780): a task link with a trajectory gets its
 * f* segment from the quintic / slerp trajectory + PD law on the device; link_index = position of the link inside its level.
 * TaskLink::SetTaskGain(pos_p, pos_d, pos_a, rot_p, rot_d, rot_a)  include/dwbc_task.h:108 (shared by the batch) */
int dwbc_batch_set_task_gain(dwbc_batch *b, int level, int link_index, const double pos_p[3], const double pos_d[3],
                             const double pos_a[3], const double rot_p[3], const double rot_d[3], const double rot_a[3]);
/* TaskLink::SetTrajectoryQuintic + SetTrajectoryRotation  include/dwbc_task.h:104-106 : traj is B x 34 doubles per instance
 *   t_start t_end | pos_init[3] vel_init[3] pos_desired[3] vel_desired[3] | rot_init[9] rot_desired[9] (row-major) |
 *   has_pos has_rot (traj_pos_set / traj_rot_set);  NULL = back to the SetTaskSpace values for this link */
int dwbc_batch_set_trajectory(dwbc_batch *b, int level, int link_index, const double *traj);
/* RobotData::control_time_ (include/dwbc.h:86), one value per instance */
int dwbc_batch_set_control_time(dwbc_batch *b, const double *control_time);
int dwbc_batch_set_torque_limit(dwbc_batch *b, const double *tau_lim);      /* SetTorqueLimit include/dwbc.h:249; NULL = unset */
int dwbc_batch_fstar_size(const dwbc_batch *b);
int dwbc_batch_task_dof(const dwbc_batch *b, int level);

/* UpdateKinematics(q, qdot, qddot) include/dwbc.h:251 : q is B x (ndof+1); qdot (B x ndof) may be NULL: only B_, the link
 * velocities and the on-device task reference read it; qddot is accepted for signature parity and unused */
int dwbc_batch_set_state(dwbc_batch *b, const double *q, const double *qdot, const double *qddot);
/* SetContact(bool...) include/dwbc.h:291 : flags is B x n_contacts */
int dwbc_batch_set_contact(dwbc_batch *b, const uint8_t *flags);
/* SetContact with a third flag raised: the reference stacks every flagged contact (src/dwbc.cpp:445-453; its tests register both
 * hands next to the feet, tests/dwbc_test.cpp:68-69).  The product kernels stack two; n = 3 routes every solve of this batch through
 * the general-contact kernel (up to three simultaneously active 6D contacts per instance, hqp = true, link tasks with f* from
 * SetTaskSpace) and widens DWBC_WRENCH to 6 n doubles per instance.  Call before binding a wrench buffer.  n = 2 restores the default. */
int dwbc_batch_set_max_active_contacts(dwbc_batch *b, int n);
int dwbc_batch_max_active_contacts(const dwbc_batch *b);
/* SetTaskSpace(level, f*) include/dwbc.h:333 : fstar is B x task_dof(level) */
int dwbc_batch_set_fstar(dwbc_batch *b, int level, const double *fstar);

/* CopyKinematicsData(target) include/dwbc.h:375, src/dwbc.cpp:1711-1762: state, contact flags, task spaces with their f*, torque
 * limit and control time of `src` into `dst` (same model and batch size); the hand-off the reference uses between threads */
int dwbc_batch_copy_kinematics(dwbc_batch *dst, const dwbc_batch *src);

/* page-locked host mirror of an input field (DWBC_IN_Q, DWBC_IN_CONTACT, DWBC_IN_FSTAR; NULL for anything else or before the
 * field has a size): a caller that assembles its states directly in this memory and passes the same pointer to
 * dwbc_batch_set_state / set_contact (or calls dwbc_batch_set_fstar with pointers into it -- level l starts at column
 * fstar offset of l) skips the host-side copy; the upload is one asynchronous PCIe transfer on the batch's stream.
 * Lifetime of the contents: dwbc_batch_solve starts that transfer and returns; the mirror may be REWRITTEN only after the next call
 * of dwbc_batch_host_ptr / dwbc_batch_set_state / set_contact / set_fstar (each waits for the pending upload first) or after
 * dwbc_batch_sync / dwbc_batch_get.  Writing through a pointer kept from before the solve without one of those calls races the DMA. */
void *dwbc_batch_host_ptr(dwbc_batch *b, int field);
/* zero-copy: use a caller-owned DEVICE buffer (e.g. a torch tensor's data_ptr) for an input or output field */
int dwbc_batch_bind_device(dwbc_batch *b, int field, void *device_ptr);
int dwbc_batch_set_stream(dwbc_batch *b, void *hip_stream);
int dwbc_batch_enable_dump(dwbc_batch *b, int on);

/* CalcContactConstraint + CalcGravCompensation + CalcTaskControlTorque(hqp,init) + CalcContactRedistribute(hqp,init)
 * (include/dwbc.h:280,246,349,298) as ONE fused kernel launch on the batch's stream (asynchronous). */
int dwbc_batch_solve(dwbc_batch *b, unsigned flags);
int dwbc_batch_sync(dwbc_batch *b);
/* K back-to-back solves bracketed by HIP events on the batch's stream; returns total milliseconds in *ms */
int dwbc_batch_time_solves(dwbc_batch *b, unsigned flags, int steps, float *ms);
/* copy a field of all B instances to host memory (synchronises the stream) */
int dwbc_batch_get(dwbc_batch *b, int field, void *host_out, size_t bytes);
size_t dwbc_batch_field_bytes(const dwbc_batch *b, int field);
/* kernel launch geometry, for DESIGN.md / bench bookkeeping */
int dwbc_batch_launch_info(const dwbc_batch *b, int *threads_per_instance, int *lds_bytes);
/* name of the kernel the next dwbc_batch_solve will launch (as rocprofv3 prints it), for bench / profile bookkeeping */
const char *dwbc_batch_kernel_name(const dwbc_batch *b);

/* ---- generic hierarchical-QP class for a batch: DWBC::HQP / HQP_Hierarch (reference include/dwbc_hqp.h:8-141,
 * src/dwbc_hqp.cpp).  Every level i poses  A_i y + a_i <= v (inequalities with slack), B_i y + b_i = w (equalities, least
 * squares), optional cost 1/2 y^T H y; levels are solved in sequence inside the null space of the earlier equalities.  The
 * reference solves each level with OSQP (not vendored); here each level is solved exactly on the device (DESIGN.md "HQP
 * class").  All matrices are per instance, batch-major, row-major: A is B x ineq x nv, nv = acceleration + torque + contact. */
typedef struct dwbc_hqp dwbc_hqp;
enum dwbc_hqp_field {
    DWBC_HQP_Y_ANS = 0,   /* (nv)   f64  hqp_hs_[level].y_ans_ */
    DWBC_HQP_V_ANS = 1,   /* (ineq) f64  hqp_hs_[level].v_ans_ */
    DWBC_HQP_W_ANS = 2,   /* (eq)   f64  hqp_hs_[level].w_ans_ = B y + b */
    DWBC_HQP_STATUS = 3,  /* (1)    i32  1 solved / 0 failed (iteration limit, working set overflow, singular Hessian) */
    DWBC_HQP_ITER = 4,    /* (1)    i32  active-set steps of the level */
    DWBC_HQP_NULL_SIZE = 5, /* (1)  i32  hqp_hs_[level].null_space_size_ */
    DWBC_HQP_A = 6, DWBC_HQP_a = 7, DWBC_HQP_B = 8, DWBC_HQP_b = 9  /* the (normalised) level matrices as stored */
};
dwbc_hqp *dwbc_hqp_create(int B, int device, int acceleration_size, int torque_size, int contact_size); /* HQP::initialize :16-21 */
void dwbc_hqp_destroy(dwbc_hqp *h);
int dwbc_hqp_add_hierarchy(dwbc_hqp *h, int ineq_const_size, int eq_const_size);   /* HQP::addHierarchy :425-434; returns the level */
int dwbc_hqp_clear(dwbc_hqp *h);                                                    /* drop every level */
/* HQP_Hierarch::updateConstraintMatrix :530-547 (A, a may be NULL for a level without inequalities) */
int dwbc_hqp_update_constraint_matrix(dwbc_hqp *h, int level, const double *A, const double *a, const double *Bm, const double *b);
int dwbc_hqp_update_cost_matrix(dwbc_hqp *h, int level, const double *H, const double *g);  /* updateCostMatrix :483-493 (g is stored, never read: as in the reference) */
int dwbc_hqp_normalize_constraint_matrix(dwbc_hqp *h, int level);                   /* normalizeConstraintMatrix :555-581 */
/* HQP_Hierarch::updateInequalityCostWeight / updateEqualityCostWeight / updateConstraintWeight :503-553.  V: B x ineq x ineq, W: B x eq x eq
 * (row major per instance; NULL = identity).  As in the reference they enter dwbc_hqp_solve_first only (level 0 posed on V A, V a, W B,
 * W b, :245-254); solveSequential reads the unweighted matrices, and the weights of later levels are stored and never read. */
int dwbc_hqp_update_constraint_weight(dwbc_hqp *h, int level, const double *V, const double *W);
/* seed hqp_hs_[level].y_ans_ / v_ans_ directly, as ConfigureLQP does for level 0 (src/dwbc.cpp:4378-4381) */
int dwbc_hqp_set_answer(dwbc_hqp *h, int level, const double *y_ans, const double *v_ans);
int dwbc_hqp_prepare(dwbc_hqp *h);                    /* HQP::prepare :23-85 (the null-space chain itself is evaluated with the solve) */
int dwbc_hqp_solve_first(dwbc_hqp *h, int init);      /* HQP::solvefirst :222-289 */
int dwbc_hqp_solve_sequential(dwbc_hqp *h, int init); /* HQP::solveSequential :397-403; init is accepted and unused (exact solves have no warm state) */
int dwbc_hqp_num_levels(const dwbc_hqp *h);
size_t dwbc_hqp_field_bytes(const dwbc_hqp *h, int level, int field);
int dwbc_hqp_get(dwbc_hqp *h, int level, int field, void *host_out, size_t bytes);
/* RobotData::ConfigureLQP(hqp) src/dwbc.cpp:4304-4430 on the device, from the batch's last solve (needs dwbc_batch_enable_dump and
 * the same contact flags in every instance): (re)builds the levels of `h` -- torque limit / floating-base dynamics,
 * contact cones + acceleration limit / contact constraint, one equality level per task space -- and seeds level 0.
 * RobotData::CalcControlTorqueLQP(hqp) src/dwbc.cpp:4432-4452 is dwbc_hqp_solve_sequential(h, init). */
int dwbc_batch_configure_lqp(dwbc_batch *b, dwbc_hqp *h);
/* torque of the LQP answer, tau = A[6:] qddot + J_C^T[6:] f_c + B_[6:] (tests/sp_test/jacc_compare.cpp:416-418): B x m */
int dwbc_batch_lqp_torque(dwbc_batch *b, dwbc_hqp *h, double *tau);
/* RobotData::CalcSingleTaskTorqueWithJACC_QP(ts_[level], init) src/dwbc.cpp:3772-3945: the joint-acceleration QP of one task level
 * (variables qddot, tau, f_c and the task slack; dynamics, contact and the earlier levels' tasks as equalities; contact cones,
 * |qddot_joint| <= 10, |tau| <= 200) from the batch's last solve (dump on, one contact state per batch), solved exactly on `h`
 * (which is rebuilt for it).  Levels must be solved in order: level i uses f*_qp of the levels before it.  Results per instance:
 * ts_[level].acc_qp_ (ndof), torque_qp_ (m), contact_qp_ (12, zero padded), f_star_qp_ (6, zero padded), status (i32) */
enum dwbc_jacc_field { DWBC_JACC_ACC = 0, DWBC_JACC_TORQUE = 1, DWBC_JACC_CONTACT = 2, DWBC_JACC_FSTAR_QP = 3, DWBC_JACC_STATUS = 4 };
int dwbc_batch_solve_jacc(dwbc_batch *b, dwbc_hqp *h, int level);
int dwbc_batch_get_jacc(dwbc_batch *b, int level, int field, void *host_out, size_t bytes);

/* ---- the same formulations on the REDUCED system, after a cycle solved with DWBC_SOLVE_REDUCED (dump on, one contact state per
 * batch).  RS = vc_dof + 6 (24 for TOCABI double support): contact-chain coordinates + 6 centroidal coordinates of the other bodies.
 * The contact-chain task levels (`!noncont_task`) are taken in level order; `level` below counts THEM.
 *   RobotData::ConfigureLQP_R(hqp)            src/dwbc.cpp:4504-4632  (A_R, J_CR, G_R, J_task J_R_INV_T^T; cost scaled by |A_|_F of
 *                                              the full model; torque limit 200, 600 on row RS-6-4).  HQP sizes (RS, 0, contact dof).
 *   RobotData::CalcControlTorqueLQP_R(hqp)    :4455-4477 = dwbc_hqp_solve_sequential.  dwbc_batch_lqp_torque then returns
 *                                              B x (RS-6): the 12 / 6 chain torques and the wrench on the centroidal coordinates.
 *   CalcSingleTaskTorqueWithJACC_QP_R         :3946-4122 (|tau| <= 200 on the chain joints only); results through dwbc_batch_get_jacc
 *                                              with acc_qp_ (RS) and torque_qp_ (RS-6).
 * The non-contact halves take ONE 6-D task level on a non-contact link (the reference reads ts_[1] as such); `level` is the task level
 * as registered.  HQP sizes (nc_dof, 0, 0), nc_dof = ndof - vc_dof.
 *   RobotData::ConfigureLQP_R_NC(hqp_nc, q_acc) :4634-4760: q_acc = the last level's answer of the solved reduced LQP `hr`.  Then
 *                                              CalcControlTorqueLQP_R_NC :4479-4502 = dwbc_hqp_solve_first + dwbc_hqp_solve_sequential.
 *   CalcSingleTaskTorqueWithJACC_QP_R_NC(ts, prev_acc) :4124-4302: prev_acc = acc_qp_ of the reduced JACC level `src_level`.  Results
 *                                              (dwbc_batch_get_jacc_nc): acc_qp_ (nc_dof), torque_qp_ (nc_dof), gacc_qp_ (6, under
 *                                              DWBC_JACC_CONTACT), f_star_qp_ (6), status. */
/* vc_dof and nc_dof of the contact state set through dwbc_batch_set_contact (src/dwbc.cpp:2818-2823); 0 if the contact chains
 * do not occupy the leading joint dofs (the reduced path's scope) */
int dwbc_batch_reduced_dims(dwbc_batch *b, int *vc_dof, int *nc_dof);
int dwbc_batch_configure_lqp_r(dwbc_batch *b, dwbc_hqp *h);
int dwbc_batch_configure_lqp_r_nc(dwbc_batch *b, dwbc_hqp *h_nc, const dwbc_hqp *hr, int level);
int dwbc_batch_solve_jacc_r(dwbc_batch *b, dwbc_hqp *h, int level);
int dwbc_batch_solve_jacc_r_nc(dwbc_batch *b, dwbc_hqp *h_nc, int level, int src_level);
int dwbc_batch_get_jacc_nc(dwbc_batch *b, int field, void *host_out, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
