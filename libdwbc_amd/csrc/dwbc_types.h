// dwbc_types.h -- POD structures shared by the host C-ABI layer and the HIP kernels.
// MI355X-native batched libdwbc hot path; see DESIGN.md.  No torch / Eigen types here.
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

// arithmetic type of the device path.  The product library is built for double; DWBC_REAL=float (with the namespace renamed
// by -Ddwbc=dwbc_f32) builds the fp32 variant of the same kernels in a second translation unit (dwbc_kernels_f32.hip).
#ifndef DWBC_REAL
#define DWBC_REAL double
#endif

namespace dwbc {

typedef DWBC_REAL real_t;
typedef double io_t;  // element type of every state / command / torque buffer at the boundary, whatever real_t is
constexpr bool kF32 = sizeof(real_t) == 4;

constexpr int kMaxBodies = 48;
constexpr int kMaxContacts = 4;       // registered contacts (reference tests register 4: tests/dwbc_test.cpp:66-69)
constexpr int kMaxActiveContacts = 2; // simultaneously active 6D contacts the fused kernel is sized for
constexpr int kMaxLevels = 4;
constexpr int kMaxTaskLinks = 2;
constexpr int kMaxTaskDof = 6;        // per level on the product kernels (one 6D link or two 3-dof links)
constexpr int kMaxTaskDofWide = 12;   // per level through the general-contact kernel: two 6D links on one level (both hands: reference
                                      // tests/sp_test/regulation_test.cpp:90-91, src/dwbc.cpp:592-600)
constexpr int kMaxReducedDof = 24;    // reduced system: 6 + 12 contact-chain joints of two legs + 6 centroidal coordinates
constexpr int kBodyStride = 25;       // doubles per body in the device model table

// per-body record (doubles): R_T[9] p_T[3] axis[3] mass com[3] Icom[6]{xx,xy,xz,yy,yz,zz}
enum BodyField { BF_RT = 0, BF_PT = 9, BF_AXIS = 12, BF_MASS = 15, BF_COM = 16, BF_ICOM = 19 };

// task link modes -- same numbering as reference include/dwbc_task.h:23-33
enum TaskMode {
    TASK_LINK_6D = 0, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME,
    TASK_LINK_POSITION, TASK_LINK_POSITION_COM_FRAME, TASK_LINK_POSITION_CUSTOM_FRAME,
    TASK_LINK_ROTATION, TASK_LINK_ROTATION_CUSTOM_FRAME
};

// Problem setup shared by every instance of a batch (kernel argument, read through scalar loads).
struct Setup {
    int nb, ndof, maxdepth;
    int n_contacts;
    int c_link[kMaxContacts];
    double c_point[kMaxContacts][3];
    double c_lx[kMaxContacts], c_ly[kMaxContacts], c_mu[kMaxContacts], c_muz[kMaxContacts];
    int n_levels;
    int t_nlinks[kMaxLevels];
    int t_mode[kMaxLevels][kMaxTaskLinks];
    int t_link[kMaxLevels][kMaxTaskLinks];
    double t_point[kMaxLevels][kMaxTaskLinks][3];
    int t_dof[kMaxLevels];
    int fstar_off[kMaxLevels];
    int fstar_total;
    int has_tau_lim;
    double tau_lim[48];
    int qp_max_iter_task;   // 1000, reference src/dwbc.cpp:1080
    int qp_max_iter_contact; // 300,  reference src/dwbc.cpp:1546
    // structural support of the Jacobians: bit d set <=> dof d (0..5 base, 6.. joints) moves the link.  Derived from the
    // kinematic tree when a contact / task link is registered; lets the products J A^-1 and J A^-1 N_c skip zero columns.
    int parent[kMaxBodies];
    int topo_kind;  // 1: the parent table equals TopoTocabi's (dwbc_topo.h) and the kernels may use its constant sparsity
    unsigned long long c_dofmask[kMaxContacts];
    unsigned long long t_dofmask[kMaxLevels];
    // on-device task reference (dwbc_fstar.h): gains of TaskLink::SetTaskGain (pos_p pos_d pos_a rot_p rot_d, 3 each) and the
    // trajectory slot of every task link (-1: f* comes from SetTaskSpace)
    double t_gain[kMaxLevels][kMaxTaskLinks][15];
    int t_traj_slot[kMaxLevels][kMaxTaskLinks];
    int n_traj;
    int has_com_task;  // some task level controls the synthetic COM link (link id = nb)
    // TASK_CUSTOM levels (reference include/dwbc.h:318,333): the caller supplies J_task per instance; slot into BatchIO::custom_J or -1
    int t_custom_slot[kMaxLevels];
    int n_custom;
};

// per-instance diagnostics (int32)
enum DiagField {
    DG_ST_CONTACT = 0, DG_ST_TASK, DG_ST_REDIS, DG_FAIL_LEVEL,
    DG_QP_ITER = 4,  // [kMaxLevels+1]
    DG_QP_NACT = 9,  // [kMaxLevels+1]
    DG_QP_ACT = 14,  // [kMaxLevels+1][12]
    DG_TIME = 14 + 5 * 12,  // [16] stage stamps (shader cycles since kernel start), only in the DWBC_STAGE_TIMERS build
#ifdef DWBC_STAGE_TIMERS
    DG_FTIME = 14 + 5 * 12 + 16,  // [64] fine-grained stamps of the diagnostic build (lean and full kernels alike)
    DG_COUNT = 14 + 5 * 12 + 16 + 64
#else
    DG_COUNT = 14 + 5 * 12 + 16
#endif
};

// dump layout (doubles per instance) for the debug / facade getters; N = ndof, M = N-6, C = 12
struct DumpLayout {
    int N, M;
    int A, A_inv, J_C, Lambda_c, J_C_INV_T, A_inv_N_C, W_inv, NwJw, Vb, G, P_C, link_R, link_p;
    int J_task, Lambda_task, J_kt, X, Y, fstar_qp, contact_qp, cf_redis, qp_viol, CMM, com, com_inertia, J_com, B, link_v, link_w, contact_pos, contact_rot, zmp, A_R_inv, A_R, G_R, J_I_nc, J_I_nc_inv_T, total;
    __host__ __device__ static DumpLayout make(int n) {
        DumpLayout d;
        d.N = n;
        d.M = n - 6;
        int o = 0;
        const int C = 6 * kMaxActiveContacts, K = C - 6, T = kMaxTaskDof, L = kMaxLevels;
        d.A = o; o += n * n;
        d.A_inv = o; o += n * n;
        d.J_C = o; o += C * n;
        d.Lambda_c = o; o += C * C;
        d.J_C_INV_T = o; o += C * n;
        d.A_inv_N_C = o; o += n * n;
        d.W_inv = o; o += d.M * d.M;
        d.NwJw = o; o += d.M * K;
        d.Vb = o; o += d.M * K;
        d.G = o; o += n;
        d.P_C = o; o += C;
        d.link_R = o; o += kMaxBodies * 9;
        d.link_p = o; o += kMaxBodies * 3;
        d.J_task = o; o += L * T * n;
        d.Lambda_task = o; o += L * T * T;
        d.J_kt = o; o += L * d.M * T;
        d.X = o; o += L * d.M * T;
        d.Y = o; o += L * T * d.M;
        d.fstar_qp = o; o += L * T;
        d.contact_qp = o; o += L * K;
        d.cf_redis = o; o += K;
        d.qp_viol = o; o += L + 1;
        d.CMM = o; o += 6 * n;          // CMM_ (dwbc.cpp:336)
        d.com = o; o += 3;              // com_pos (dwbc.cpp:322)
        d.com_inertia = o; o += 9;      // link_.back().inertia (dwbc.cpp:343)
        d.J_com = o; o += 6 * n;        // link_.back().jac_com_ (dwbc.cpp:352)
        d.B = o; o += n;                // B_ (dwbc.cpp:343-344), needs qdot
        d.link_v = o; o += kMaxBodies * 3;  // link_[i].v (link.cpp:87), needs qdot
        d.link_w = o; o += kMaxBodies * 3;  // link_[i].w (link.cpp:88)
        d.contact_pos = o; o += kMaxActiveContacts * 3;  // cc_[i].xc_pos of the active contacts (contact_constraint.cpp:53)
        d.contact_rot = o; o += kMaxActiveContacts * 9;  // cc_[i].rotm
        d.zmp = o; o += 3 + kMaxActiveContacts * 3;      // getZMP(getContactForce(tau_total)), then cc_[i].zmp_pos (dwbc.cpp:898-939)
        // reduced (centroidal) model, written by the reduced cycle only (dwbc.cpp:2932-2988); RS <= kMaxReducedDof, row stride kMaxReducedDof
        d.A_R_inv = o; o += kMaxReducedDof * kMaxReducedDof;
        d.A_R = o; o += kMaxReducedDof * kMaxReducedDof;
        d.G_R = o; o += kMaxReducedDof;
        d.J_I_nc = o; o += 6 * (n - 12);          // 6 x nc_dof, row stride n - 12
        d.J_I_nc_inv_T = o; o += 6 * (n - 12);    // 6 x nc_dof, row stride n - 12
        d.total = o;
        return d;
    }
};

struct BatchIO {  // device pointers.  Inputs and outputs are io_t (double) in both builds; the model table and the dump are real_t
    int B;
    const io_t *q;             // B x (N+1)   [x y z qx qy qz joints... qw]  (reference include/dwbc.h:251)
    const io_t *qdot;          // B x N or nullptr: [v_world(3) w_body(3) joint rates]; only B_, link velocities and the task reference use it
    const unsigned char *flags;  // B x n_contacts
    const io_t *fstar;         // B x fstar_total
    const io_t *traj;          // B x n_traj x 34 trajectory records (dwbc_fstar.h) or nullptr
    const io_t *ctime;         // B control times (RobotData::control_time_) or nullptr
    const io_t *custom_J;      // B x n_custom x (kMaxTaskDof x N) row-major J_task of the TASK_CUSTOM levels, or nullptr
    io_t *tau;                 // B x 3 x M : torque_grav_, torque_task_, torque_contact_
    io_t *wrench;              // B x 12    : getContactForce(tau_total), zero padded
    int *status;                 // B         : 1 ok / 0 fail (reference int returns ANDed)
    int *diag;                   // B x DG_COUNT
    real_t *dump;                // B x DumpLayout::total or nullptr
    const real_t *body;          // nb x kBodyStride
    const int *topo;             // parent[nb], depth[nb], subtree[nb]
    int hqp;                     // 1: CalcTaskControlTorque / CalcContactRedistribute with hqp = true (QPs); 0: plain hierarchy +
                                 // closed-form redistribution (dwbc_nohqp.h)
    int pair_swap_bit;           // paired kernel (dwbc_cycle2p.h): workgroups with this bit of their index set swap the roles of their two
                                 // waves (-1: never)
    int warm;                    // 1: init = false (reference src/dwbc.cpp:1064-1074, src/qp_wrapper.cpp:249-296): every QP first tries
                                 // the rows of its previous working set (diag[DG_QP_ACT..], written by the previous launch)
    int wrench_ld;               // general-contact kernel (dwbc_cycle_gc.h): doubles per instance in `wrench` (18 for three contacts); the
                                 // product kernels write 12 per instance and do not read it
};

}  // namespace dwbc
