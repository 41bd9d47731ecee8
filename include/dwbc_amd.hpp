// dwbc_amd.hpp -- header-only C++ facade that gives libdwbc's RobotData names and call sequence back on top of the
// C-ABI (include/dwbc_batch.h, libdwbc_hip.so).  One object = one robot (a batch of 1) so that an existing libdwbc
// control loop (reference example/main.cpp:62-108, tests/dwbc_test.cpp:61-130) compiles against it with
//     using DWBC::RobotData;     // from this header instead of the reference's dwbc.h
// Differences that are deliberate and visible:
//   * vectors / matrices are dwbc_amd::Vec / dwbc_amd::Mat (std::vector<double> based, row-major).  Every entry point that
//     takes a vector is also a template over any type with data() / size() (Eigen::VectorXd, std::array, ...), and
//     dwbc_amd::to_vector<V>(vec) / to_matrix<M>(mat) build the caller's own types from the results (M needs a
//     (rows, cols) constructor and operator()(i, j): Eigen::MatrixXd qualifies) -- so code written with Eigen types compiles
//     against this header without this header including Eigen;
//   * every Calc* call launches the fused kernel (CalcContactConstraint .. CalcContactRedistribute are one launch on the
//     device) and refreshes all public fields, so the reference's call order yields the reference's values; a loop that
//     only needs the final torque should call CalcAll() once per tick;
//   * only what the device path implements: floating base, CONTACT_6D, link / COM tasks (TASK_LINK_*), hqp = true or false.
//
// Reference API mirrored: include/dwbc.h:59-430 (method names, argument order, int 1/0 returns, std::cout messages).
#pragma once
#include <cstdint>
#include <cstring>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "dwbc_batch.h"

namespace dwbc_amd {
typedef std::vector<double> Vec;
struct Mat {  // row-major dense matrix
    int rows = 0, cols = 0;
    std::vector<double> d;
    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.0) {}
    double &operator()(int i, int j) { return d[(size_t)i * cols + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * cols + j]; }
};
struct Vec3 {
    double v[3];
    Vec3(double x = 0, double y = 0, double z = 0) : v{x, y, z} {}
    template <class V3, class = decltype(std::declval<const V3 &>().data()), class = decltype(std::declval<const V3 &>().size())>
    Vec3(const V3 &o) : v{o.data()[0], o.data()[1], o.data()[2]} {}  // Eigen::Vector3d and the like
    const double *data() const { return v; }
    size_t size() const { return 3; }
};
// any contiguous vector-like (data(), size()) -> Vec
template <class V>
inline Vec as_vec(const V &x) { return Vec(x.data(), x.data() + x.size()); }
inline const Vec &as_vec(const Vec &x) { return x; }
// results into the caller's own types
template <class V>
inline V to_vector(const Vec &x) { V o(x.size()); for (size_t i = 0; i < x.size(); i++) o[i] = x[i]; return o; }
template <class M>
inline M to_matrix(const Mat &x) { M o(x.rows, x.cols); for (int i = 0; i < x.rows; i++) for (int j = 0; j < x.cols; j++) o(i, j) = x(i, j); return o; }
}  // namespace dwbc_amd

namespace DWBC {

using dwbc_amd::Mat;
using dwbc_amd::Vec;
using dwbc_amd::Vec3;

enum { JOINT_FLOATING_BASE, JOINT_6DOF, JOINT_REVOLUTE, JOINT_PRISMATIC, JOINT_FIXED };  // dwbc_link.h:12-19 (AddLink's joint_type)
enum TASK_TYPE { TASK_UNDEFINED, TASK_LINK, TASK_CUSTOM, TASK_NONCONTACT_CHAIN, TASK_CONTACT_CHAIN, TASK_CENTROIDAL };  // dwbc_task.h:12-20
enum CONTACT_TYPE { CONTACT_6D = 0, CONTACT_POINT = 1, CONTACT_LINK = 2, CONTACT_LINE = 3 };  // dwbc_contact_constraint.h:19-25
enum TASK_LINK_MODE {                                                                          // dwbc_task.h:23-33
    TASK_LINK_6D = 0, TASK_LINK_6D_COM_FRAME, TASK_LINK_6D_CUSTOM_FRAME, TASK_LINK_POSITION, TASK_LINK_POSITION_COM_FRAME,
    TASK_LINK_POSITION_CUSTOM_FRAME, TASK_LINK_ROTATION, TASK_LINK_ROTATION_CUSTOM_FRAME
};

struct TaskLinkView {  // DWBC::TaskLink: trajectory + PD reference of one task link (include/dwbc_task.h:49-128)
    bool traj_pos_set = false, traj_rot_set = false, dirty = false;
    double rec[34] = {0};   // the C-ABI trajectory record (include/dwbc_batch.h, dwbc_batch_set_trajectory)
    double gain[18] = {0};  // pos_p pos_d pos_a rot_p rot_d rot_a
    void SetTrajectoryQuintic(double start_time, double end_time, Vec3 pos_init, Vec3 vel_init, Vec3 pos_desired, Vec3 vel_desired) {
        rec[0] = start_time; rec[1] = end_time;
        for (int a = 0; a < 3; a++) { rec[2 + a] = pos_init.v[a]; rec[5 + a] = vel_init.v[a]; rec[8 + a] = pos_desired.v[a]; rec[11 + a] = vel_desired.v[a]; }
        traj_pos_set = true; rec[32] = 1.0; dirty = true;
    }
    void SetTrajectoryRotation(double start_time, double end_time, const Mat &rot_init, Vec3 /*twist_init*/, const Mat &rot_desired, Vec3 /*twist_desired*/) {
        rec[0] = start_time; rec[1] = end_time;
        for (int a = 0; a < 9; a++) { rec[14 + a] = rot_init.d[a]; rec[23 + a] = rot_desired.d[a]; }
        traj_rot_set = true; rec[33] = 1.0; dirty = true;
    }
    void SetTaskGain(Vec3 pos_p, Vec3 pos_d, Vec3 pos_a, Vec3 rot_p, Vec3 rot_d, Vec3 rot_a) {
        for (int a = 0; a < 3; a++) { gain[a] = pos_p.v[a]; gain[3 + a] = pos_d.v[a]; gain[6 + a] = pos_a.v[a]; gain[9 + a] = rot_p.v[a]; gain[12 + a] = rot_d.v[a]; gain[15 + a] = rot_a.v[a]; }
        dirty = true;
    }
};
struct TaskSpaceView {  // the fields of DWBC::TaskSpace callers read (include/dwbc_task.h:130-171)
    int task_dof_ = 0;
    std::vector<TaskLinkView> task_link_;
    Vec f_star_, f_star_qp_, contact_qp_;
    Vec acc_qp_, torque_qp_, gacc_qp_;  // CalcSingleTaskTorqueWithJACC_QP* results (include/dwbc_task.h)
    Mat J_task_, Lambda_task_, J_kt_;
    int qp_error = 0;
};
struct LinkView {  // the kinematic fields of DWBC::Link callers read (include/dwbc_link.h; src/link.cpp:76-96)
    Vec3 xpos, v, w;
    Mat rotm;
};
struct ContactView {  // include/dwbc_contact_constraint.h:27-80
    int link_number_ = -1;
    Vec3 xc_pos, zmp_pos;  // include/dwbc_contact_constraint.h: contact point (world) and the contact's ZMP
    Mat rotm;
    bool contact = false;
    int contact_dof_ = 6, constraint_number_ = 10;
    double contact_plane_x_ = 0.0, contact_plane_y_ = 0.0, friction_ratio_ = 0.2, friction_ratio_z_ = 0.2;  // dwbc_contact_constraint.h:44-52
};

// DWBC::HQP / HQP_Hierarch (include/dwbc_hqp.h:8-141): the fields callers read after a solve, one instance
struct HQP_Hierarch {
    int ineq_const_size_ = 0, eq_const_size_ = 0, null_space_size_ = 0, variable_size_ = 0;
    int qp_status_ = 1, qp_iter_ = 0;
    Vec y_ans_, v_ans_, w_ans_;
};
class RobotData;
class HQP {
  public:
    int acceleration_size_ = 0, torque_size_ = 0, contact_size_ = 0;
    std::vector<HQP_Hierarch> hqp_hs_;
    HQP() {}
    ~HQP() { if (h_) dwbc_hqp_destroy(h_); }
    HQP(const HQP &) = delete;
    HQP &operator=(const HQP &) = delete;
    void initialize(int acceleration_size, int torque_size, int contact_size, int device = 0) {  // dwbc_hqp.cpp:16-21
        if (h_) dwbc_hqp_destroy(h_);
        acceleration_size_ = acceleration_size; torque_size_ = torque_size; contact_size_ = contact_size;
        h_ = dwbc_hqp_create(1, device, acceleration_size, torque_size, contact_size);
        if (!h_) std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl;
        hqp_hs_.clear();
    }
    void addHierarchy(int ineq_const_size, int eq_const_size) {  // dwbc_hqp.cpp:425-434
        if (dwbc_hqp_add_hierarchy(h_, ineq_const_size, eq_const_size) < 0) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        HQP_Hierarch hh;
        hh.ineq_const_size_ = ineq_const_size; hh.eq_const_size_ = eq_const_size;
        hh.variable_size_ = acceleration_size_ + torque_size_ + contact_size_;
        hqp_hs_.push_back(hh);
    }
    // hqp_hs_[level].updateConstraintMatrix / updateCostMatrix / normalizeConstraintMatrix (dwbc_hqp.cpp:483-581)
    void updateConstraintMatrix(int level, const Mat &A, const Vec &a, const Mat &B, const Vec &b) {
        check(dwbc_hqp_update_constraint_matrix(h_, level, A.d.empty() ? nullptr : A.d.data(), a.empty() ? nullptr : a.data(), B.d.empty() ? nullptr : B.d.data(), b.empty() ? nullptr : b.data()));
    }
    void updateCostMatrix(int level, const Mat &H, const Vec &g) { check(dwbc_hqp_update_cost_matrix(h_, level, H.d.data(), g.data())); }
    void normalizeConstraintMatrix(int level) { check(dwbc_hqp_normalize_constraint_matrix(h_, level)); }
    // hqp_hs_[level].updateInequalityCostWeight / updateEqualityCostWeight / updateConstraintWeight (dwbc_hqp.cpp:503-553): full matrices,
    // or vectors taken as diagonals (the VectorXd overloads); read by solvefirst only, as in the reference (:245-254)
    void updateConstraintWeight(int level, const Mat &V, const Mat &W) {
        check(dwbc_hqp_update_constraint_weight(h_, level, V.d.empty() ? nullptr : V.d.data(), W.d.empty() ? nullptr : W.d.data()));
        if ((int)wV_.size() <= level) { wV_.resize(level + 1); wW_.resize(level + 1); }
        wV_[level] = V; wW_[level] = W;
    }
    void updateInequalityCostWeight(int level, const Mat &V) { updateConstraintWeight(level, V, kept(wW_, level)); }
    void updateEqualityCostWeight(int level, const Mat &W) { updateConstraintWeight(level, kept(wV_, level), W); }
    void updateInequalityCostWeight(int level, const Vec &V) { updateInequalityCostWeight(level, diag(V)); }
    void updateEqualityCostWeight(int level, const Vec &W) { updateEqualityCostWeight(level, diag(W)); }
    void prepare(bool = false) { check(dwbc_hqp_prepare(h_)); }  // dwbc_hqp.cpp:23-85
    void solvefirst(bool init = true) { if (check(dwbc_hqp_solve_first(h_, init))) fetch(); }       // dwbc_hqp.cpp:222-289
    void solveSequential(bool init = true, bool = false) { if (check(dwbc_hqp_solve_sequential(h_, init))) fetch(); }  // :397-403
    dwbc_hqp *handle() { return h_; }
    void fetch() {  // answers of every level into hqp_hs_
        const int n = dwbc_hqp_num_levels(h_);
        if ((int)hqp_hs_.size() != n) hqp_hs_.resize(n);
        for (int lv = 0; lv < n; lv++) {
            HQP_Hierarch &hh = hqp_hs_[lv];
            hh.variable_size_ = acceleration_size_ + torque_size_ + contact_size_;
            hh.ineq_const_size_ = (int)(dwbc_hqp_field_bytes(h_, lv, DWBC_HQP_V_ANS) / 8);
            hh.eq_const_size_ = (int)(dwbc_hqp_field_bytes(h_, lv, DWBC_HQP_W_ANS) / 8);
            hh.y_ans_.assign(hh.variable_size_, 0.0); hh.v_ans_.assign(hh.ineq_const_size_, 0.0); hh.w_ans_.assign(hh.eq_const_size_, 0.0);
            dwbc_hqp_get(h_, lv, DWBC_HQP_Y_ANS, hh.y_ans_.data(), hh.y_ans_.size() * 8);
            if (hh.ineq_const_size_) dwbc_hqp_get(h_, lv, DWBC_HQP_V_ANS, hh.v_ans_.data(), hh.v_ans_.size() * 8);
            if (hh.eq_const_size_) dwbc_hqp_get(h_, lv, DWBC_HQP_W_ANS, hh.w_ans_.data(), hh.w_ans_.size() * 8);
            dwbc_hqp_get(h_, lv, DWBC_HQP_STATUS, &hh.qp_status_, sizeof(int));
            dwbc_hqp_get(h_, lv, DWBC_HQP_ITER, &hh.qp_iter_, sizeof(int));
            dwbc_hqp_get(h_, lv, DWBC_HQP_NULL_SIZE, &hh.null_space_size_, sizeof(int));
        }
    }

  private:
    dwbc_hqp *h_ = nullptr;
    std::vector<Mat> wV_, wW_;  // the weights handed over so far (one of the two setters keeps the other's matrix)
    static Mat kept(const std::vector<Mat> &v, int level) { return level < (int)v.size() ? v[level] : Mat(); }
    static Mat diag(const Vec &v) { Mat m((int)v.size(), (int)v.size()); for (int i = 0; i < (int)v.size(); i++) m(i, i) = v[i]; return m; }
    bool check(int ok) { if (!ok) std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return ok != 0; }
};

class RobotData {
  public:
    unsigned int system_dof_ = 0, model_dof_ = 0, contact_dof_ = 0, contact_link_num_ = 0, link_num_ = 0;
    bool is_floating_ = true;
    double total_mass_ = 0.0;
    double control_time_ = 0.0;  // include/dwbc.h:86: time fed to the task-link trajectories
    Vec q_system_, q_dot_system_, q_ddot_system_;
    Vec G_, B_, torque_grav_, torque_task_, torque_contact_, torque_limit_;  // B_ = C qdot + g (dwbc.cpp:343-344)
    std::vector<LinkView> link_;  // link_[i].xpos / rotm / v / w (src/link.cpp:78-88)
    Mat A_, A_inv_, J_C, Lambda_contact, J_C_INV_T, N_C, A_inv_N_C, W, W_inv, NwJw;
    Mat CMM_, J_com_, com_inertia_;  // include/dwbc.h:114, link_.back().jac_com_, link_.back().inertia
    Vec3 com_pos;                    // include/dwbc.h:141
    Vec P_C, cf_redis_qp_;
    // reduced (centroidal) model, filled after ReducedDynamicsCalculate() (include/dwbc.h:150-200; src/dwbc.cpp:2818-2988)
    unsigned int vc_dof = 0, nc_dof = 0, co_dof = 0, reduced_model_dof_ = 0, reduced_system_dof_ = 0;
    Mat A_R, A_R_inv, J_I_nc_, J_I_nc_inv_T;
    Vec G_R;
    bool torque_limit_set_ = false;
    std::vector<TaskSpaceView> ts_;
    std::vector<ContactView> cc_;

    RobotData() {}
    ~RobotData() { release(); }
    RobotData(const RobotData &) = delete;
    RobotData &operator=(const RobotData &) = delete;

    // ---- RobotData::LoadModelData (dwbc.h:237)
    void LoadModelData(std::string urdf_path, bool floating, int verbose = 0, int device = 0) {
        release();
        device_ = device;
        adopt_model(dwbc_model_create_from_urdf(urdf_path.c_str(), floating ? 1 : 0), floating, verbose);
    }
    // ---- init-time model surgery (dwbc.h:206-226; src/dwbc.cpp:1764-2382, 2707-2730), BEFORE contacts and task spaces are added
    //      (link ids move).  The edited model needs the kernel pack of its size (`make -C libdwbc_amd/csrc pack N=.. NB=..`).
    void DeleteLink(std::string link_name, bool verbose = false) { DeleteLink(getLinkID(link_name), verbose); }
    void DeleteLink(int link_idx, bool verbose = false) { if (model_) replace_model(dwbc_model_delete_link(model_, link_idx), verbose); }
    void AddLink(int parent_link_id, const char *link_name, int joint_type, const Vec3 &joint_axis, const Mat &joint_rotm, const Vec3 &joint_trans, double body_mass,
                 const Vec3 &com_position, const Mat &inertia, bool verbose = false) {
        if (model_) replace_model(dwbc_model_add_link(model_, parent_link_id, link_name, joint_type == JOINT_FIXED ? 0 : (joint_type == JOINT_REVOLUTE ? 1 : -1), joint_axis.v,
                                                      joint_rotm.d.data(), joint_trans.v, body_mass, com_position.v, inertia.d.data()), verbose);
    }
    void ChangeLinkToFixedJoint(std::string link_name, bool verbose = false) { if (model_) replace_model(dwbc_model_change_link_to_fixed_joint(model_, getLinkID(link_name)), verbose); }
    void ChangeLinkInertia(std::string link_name, const Mat &com_inertia, const Vec3 &com_position, double com_mass, bool verbose = false) {
        if (model_) replace_model(dwbc_model_change_link_inertia(model_, getLinkID(link_name), com_inertia.d.data(), com_position.v, com_mass), verbose);
    }
    void adopt_model(dwbc_model *m, bool floating, int verbose) {
        model_ = m;
        if (!model_) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        batch_ = dwbc_batch_create(model_, 1, device_, DWBC_F64);
        if (!batch_) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        dwbc_batch_enable_dump(batch_, 1);
        is_floating_ = floating;
        system_dof_ = dwbc_model_system_dof(model_);
        model_dof_ = system_dof_ - 6;
        link_num_ = dwbc_model_num_links(model_);
        total_mass_ = dwbc_model_total_mass(model_);
        q_system_.assign(system_dof_ + 1, 0.0);
        q_system_[system_dof_] = 1.0;
        torque_grav_.assign(model_dof_, 0.0);
        torque_task_.assign(model_dof_, 0.0);
        torque_contact_.assign(model_dof_, 0.0);
        if (verbose) std::cout << "System DOF : " << system_dof_ << "  Total Mass : " << total_mass_ << std::endl;
    }
    int getLinkID(std::string link_name) { return model_ ? dwbc_model_link_id(model_, link_name.c_str()) : -1; }

    void SetTorqueLimit(const Vec &torque_limit) {  // dwbc.h:249
        torque_limit_set_ = true;
        torque_limit_ = torque_limit;
        dwbc_batch_set_torque_limit(batch_, torque_limit.data());
    }
    void UpdateKinematics(const Vec &q_virtual, const Vec &q_dot_virtual, const Vec &q_ddot_virtual, bool = true) {  // dwbc.h:251
        if (q_virtual.size() != system_dof_ + 1) { std::cout << "q size is not matched : qsize : " << system_dof_ + 1 << " input size : " << q_virtual.size() << std::endl; return; }
        if (q_dot_virtual.size() != system_dof_) { std::cout << "q_dot size is not matched" << std::endl; return; }
        if (q_ddot_virtual.size() != system_dof_) { std::cout << "q_ddot size is not matched" << std::endl; return; }
        q_system_ = q_virtual; q_dot_system_ = q_dot_virtual; q_ddot_system_ = q_ddot_virtual;
        dwbc_batch_set_state(batch_, q_system_.data(), q_dot_system_.data(), q_ddot_system_.data());
        dirty_ = true;
    }
    template <class V, class = decltype(std::declval<const V &>().data())>
    void UpdateKinematics(const V &q_virtual, const V &q_dot_virtual, const V &q_ddot_virtual, bool update = true) {
        UpdateKinematics(dwbc_amd::as_vec(q_virtual), dwbc_amd::as_vec(q_dot_virtual), dwbc_amd::as_vec(q_ddot_virtual), update);
    }
    template <class V, class = decltype(std::declval<const V &>().data())>
    void SetTorqueLimit(const V &torque_limit) { SetTorqueLimit(dwbc_amd::as_vec(torque_limit)); }
    // ---- contacts (dwbc.h:259-291)
    void AddContactConstraint(int link_number, int contact_type, Vec3 contact_point, Vec3 /*contact_vector*/, double contact_x = 0, double contact_y = 0, bool verbose = false) {
        for (auto &c : cc_) if (c.link_number_ == link_number) { std::cout << "Contact Constraint Already Exist for Link : " << link_number << std::endl; return; }
        if (dwbc_batch_add_contact(batch_, link_number, contact_type, contact_point.data(), contact_x, contact_y, 0.2, 0.2) < 0) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        ContactView c; c.link_number_ = link_number; c.contact_plane_x_ = contact_x; c.contact_plane_y_ = contact_y; cc_.push_back(c);
        flags_.push_back(0);
        if (verbose) std::cout << "#" << (cc_.size() - 1) << " Contact Constraint Added : " << dwbc_model_link_name(model_, link_number) << std::endl;
    }
    void AddContactConstraint(const char *link_name, int contact_type, Vec3 p, Vec3 n, double cx = 0, double cy = 0, bool verbose = false) {
        int id = getLinkID(link_name);
        if (id < 0) { std::cout << "Link Name is Wrong : " << link_name << std::endl; return; }
        AddContactConstraint(id, contact_type, p, n, cx, cy, verbose);
    }
    void ClearContactConstraint() {  // dwbc.h:277
        if (batch_) dwbc_batch_clear_contacts(batch_);
        cc_.clear(); flags_.clear(); contact_dof_ = 0; contact_link_num_ = 0; dirty_ = true;
    }
    void UpdateContactConstraint() { refresh(); }  // dwbc.cpp:433-454: part of the fused launch
    // getContactConstraintMatrix() (dwbc.cpp:480-513): C = -A_const_a A_rot over the active contacts, rows [4 CoP | 6 friction] per contact
    // (GetZMPConstMatrix4x6 / GetForceConstMatrix6x6, src/wbd.cpp:59-97), acting on the world-frame contact wrench
    Mat getContactConstraintMatrix() {
        refresh();
        Mat C(contact_link_num_ * 10, contact_dof_);
        int a = 0;
        for (auto &c : cc_) {
            if (!c.contact) continue;
            const double lx = c.contact_plane_x_, ly = c.contact_plane_y_, mu = c.friction_ratio_, muz = c.friction_ratio_z_;
            const double Ac[10][6] = {{0, 0, -lx, 0, -1, 0}, {0, 0, -lx, 0, 1, 0}, {0, 0, -ly, -1, 0, 0}, {0, 0, -ly, 1, 0, 0}, {1, 0, -mu, 0, 0, 0},
                                      {-1, 0, -mu, 0, 0, 0}, {0, 1, -mu, 0, 0, 0}, {0, -1, -mu, 0, 0, 0}, {0, 0, -muz, 0, 0, 1}, {0, 0, -muz, 0, 0, -1}};
            for (int r = 0; r < 10; r++)
                for (int h = 0; h < 2; h++)
                    for (int j = 0; j < 3; j++) {
                        double v = 0.0;  // (A_const_a * blkdiag(R^T, R^T))[r][3h + j] = sum_x Ac[r][3h + x] R[j][x]
                        for (int x = 0; x < 3; x++) v += Ac[r][3 * h + x] * c.rotm(j, x);
                        C(10 * a + r, 6 * a + 3 * h + j) = -v;
                    }
            a++;
        }
        return C;
    }
    void getContactConstraintMatrix(Mat &C_) { C_ = getContactConstraintMatrix(); }
    // CalcAngularMomentumMatrix (dwbc.cpp:1633-1680): angular momentum about the COM per unit generalized velocity = the angular rows of
    // CMM_; the overload with an argument returns [angular; linear] (6 x system_dof_)
    Mat CalcAngularMomentumMatrix() {
        refresh();
        Mat H(3, system_dof_);
        for (int r = 0; r < 3; r++) for (unsigned c = 0; c < system_dof_; c++) H(r, c) = CMM_(3 + r, c);
        return H;
    }
    void CalcAngularMomentumMatrix(Mat &cmm) {
        refresh();
        cmm = Mat(6, system_dof_);
        for (int r = 0; r < 3; r++) for (unsigned c = 0; c < system_dof_; c++) { cmm(r, c) = CMM_(3 + r, c); cmm(3 + r, c) = CMM_(r, c); }
    }
    template <typename... Types>
    void SetContact(Types... args) {  // dwbc.h:432-474
        std::vector<bool> v{static_cast<bool>(args)...};
        if (cc_.size() < v.size()) { std::cout << "Contact Constraint size mismatch ! input size : " << v.size() << " contact constraint size : " << cc_.size() << std::endl; return; }
        contact_link_num_ = 0; contact_dof_ = 0;
        for (size_t i = 0; i < cc_.size(); i++) {
            bool on = i < v.size() ? v[i] : false;
            cc_[i].contact = on; flags_[i] = on ? 1 : 0;
            if (on) { contact_link_num_++; contact_dof_ += 6; }
        }
        if (contact_link_num_ > 2) {  // a third contact (src/dwbc.cpp:445-453 stacks every flagged one): the general-contact kernel
            enter_general();
            if (!dwbc_batch_set_max_active_contacts(batch_, 3)) std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl;
        }
        contact_ok_ = dwbc_batch_set_contact(batch_, flags_.data()) != 0;
        if (!contact_ok_) std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl;  // e.g. more than 3 active contacts
        dirty_ = true;
    }
    // ---- tasks (dwbc.h:318-333)
    void AddTaskSpace(int heirarchy, int task_mode, int link_number, Vec3 task_point, bool verbose = false) {
        if (!dwbc_batch_add_task(batch_, heirarchy, task_mode, link_number, task_point.data())) { std::cout << dwbc_last_error() << std::endl; return; }
        if ((int)ts_.size() <= heirarchy) ts_.resize(heirarchy + 1);
        ts_[heirarchy].task_dof_ = dwbc_batch_task_dof(batch_, heirarchy);
        if (ts_[heirarchy].task_dof_ > 6) enter_general();  // a second 6D link on the level (src/dwbc.cpp:592-600): the general-contact kernel
        ts_[heirarchy].task_link_.emplace_back();
        ts_[heirarchy].f_star_.assign(ts_[heirarchy].task_dof_, 0.0);
        if (verbose) std::cout << "#" << heirarchy << " Task Space Added : " << dwbc_model_link_name(model_, link_number) << std::endl;
    }
    void AddTaskSpace(int heirarchy, int task_mode, const char *link_name, Vec3 task_point, bool verbose = false) {
        int id = getLinkID(link_name);
        if (id < 0) { std::cout << "Link Name is not Correct" << std::endl; return; }
        AddTaskSpace(heirarchy, task_mode, id, task_point, verbose);
    }
    // a further link of the same hierarchy level (dwbc.h:326-327)
    void AddTaskLink(int heirarchy, int task_mode, int link_number, Vec3 task_point, bool verbose = false) { AddTaskSpace(heirarchy, task_mode, link_number, task_point, verbose); }
    void AddTaskLink(int heirarchy, int task_mode, const char *link_name, Vec3 task_point, bool verbose = false) { AddTaskSpace(heirarchy, task_mode, link_name, task_point, verbose); }
    // TASK_CUSTOM level: the caller hands J_task over with SetTaskSpace(h, f*, J) (dwbc.h:318,333; dwbc.cpp:664-681)
    void AddTaskSpace(int heirarchy, int /*task_mode = TASK_CUSTOM*/, int task_dof, bool verbose = false) {
        if (!dwbc_batch_add_custom_task(batch_, heirarchy, task_dof)) { std::cout << dwbc_last_error() << std::endl; return; }
        if ((int)ts_.size() <= heirarchy) ts_.resize(heirarchy + 1);
        ts_[heirarchy].task_dof_ = task_dof;
        ts_[heirarchy].f_star_.assign(task_dof, 0.0);
        if (verbose) std::cout << "#" << heirarchy << " Task Space Added : custom, dof " << task_dof << std::endl;
    }
    void SetTaskSpace(int heirarchy, const Vec &f_star, const Mat &J_task) {
        if (heirarchy >= (int)ts_.size()) { std::cout << "ERROR : task space size overflow" << std::endl; return; }
        const int t = ts_[heirarchy].task_dof_;
        if (J_task.rows != t || J_task.cols != (int)system_dof_ || (int)f_star.size() != t) { std::cout << "ERROR : custom task sizes (J_task task_dof x system_dof, f_star task_dof)" << std::endl; return; }
        std::vector<double> J((size_t)t * system_dof_, 0.0);  // task_dof x system_dof_, row-major
        for (int r = 0; r < t; r++) for (unsigned c = 0; c < system_dof_; c++) J[r * system_dof_ + c] = J_task(r, c);
        if (!dwbc_batch_set_custom_task(batch_, heirarchy, f_star.data(), J.data())) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        ts_[heirarchy].f_star_ = f_star;
        dirty_ = true;
    }
    void ClearTaskSpace() { if (batch_) dwbc_batch_clear_tasks(batch_); ts_.clear(); dirty_ = true; }  // dwbc.cpp:515-520
    void ClearQP() {}  // the QP solvers live inside the kernel: nothing to allocate or clear (dwbc.h:329-330)
    void AddQP() {}
    void UpdateTaskSpace() { refresh(); }           // dwbc.cpp:685-793: part of the fused launch
    void CalcTaskSpace(bool = true) { refresh(); }  // dwbc.cpp:795-816
    void InitModelData(int = 0) {}                  // done by LoadModelData
    void InitializeMatrix() {}
    // CopyKinematicsData(target) (dwbc.cpp:1711-1762): state, torque limit, contact constraints (with their flags) and task spaces
    // (with their f*) go to the target, which then runs its own Calc* sequence; the derived quantities are recomputed there
    void CopyKinematicsData(RobotData &target_rd) {
        if (!batch_ || !target_rd.batch_) return;
        if (!dwbc_batch_copy_kinematics(target_rd.batch_, batch_)) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        target_rd.q_system_ = q_system_; target_rd.q_dot_system_ = q_dot_system_; target_rd.q_ddot_system_ = q_ddot_system_;
        target_rd.control_time_ = control_time_; target_rd.sent_time_ = control_time_;
        target_rd.torque_limit_set_ = torque_limit_set_; target_rd.torque_limit_ = torque_limit_;
        target_rd.cc_ = cc_; target_rd.flags_ = flags_; target_rd.ts_ = ts_;
        target_rd.contact_dof_ = contact_dof_; target_rd.contact_link_num_ = contact_link_num_; target_rd.contact_ok_ = contact_ok_;
        target_rd.dirty_ = true;
    }
    void printLinkInfo() {  // dwbc.cpp:2399-2423
        for (unsigned i = 0; i < link_num_; i++) std::cout << i << " : " << dwbc_model_link_name(model_, (int)i) << std::endl;
    }
    Vec GetControlTorque(bool = false, bool = true) { return Vec(); }  // the reference returns an empty vector (dwbc.cpp:1622-1631)
    int CalcContactRedistributeR(bool hqp = true, bool init = true) { return ReducedCalcContactRedistribute(hqp, init); }  // dwbc.cpp:4762-4774
    void SetTaskSpace(int heirarchy, const Vec &f_star) {  // dwbc.h:333
        if (heirarchy >= (int)ts_.size()) { std::cout << "ERROR : task space size overflow" << std::endl; return; }
        if ((int)f_star.size() != ts_[heirarchy].task_dof_) { std::cout << "ERROR : task dof not matching! heir : " << heirarchy << " fstar size : " << f_star.size() << " task_dof : " << ts_[heirarchy].task_dof_ << std::endl; return; }
        ts_[heirarchy].f_star_ = f_star;
        ts_[heirarchy].f_star_qp_.assign(f_star.size(), 0.0);
        dwbc_batch_set_fstar(batch_, heirarchy, f_star.data());
        dirty_ = true;
    }
    template <class V, class = decltype(std::declval<const V &>().data())>
    void SetTaskSpace(int heirarchy, const V &f_star) { SetTaskSpace(heirarchy, dwbc_amd::as_vec(f_star)); }
    // ---- the cycle (dwbc.h:280, 246, 349, 298, 303)
    int CalcContactConstraint() { return refresh() ? diag_[0] : 0; }
    Vec CalcGravCompensation() { refresh(); return torque_grav_; }
    int CalcTaskControlTorque(bool hqp = true, bool init = true, bool = true) {
        if (hqp != hqp_) { hqp_ = hqp; dirty_ = true; }  // hqp = false: plain hierarchy, no QP (dwbc.cpp:856-873)
        if (!refresh(init)) return 0;
        if (!hqp) { torque_contact_.assign(model_dof_, 0.0); redistributed_ = false; return diag_[1]; }
        if (general_) { torque_contact_ = tau_contact_final_; redistributed_ = true; return diag_[1]; }  // (one launch: already redistributed)
        // torque_contact_ = NwJw * contact_qp_(last level) at this point of the reference sequence (dwbc.cpp:851)
        const int k = contact_dof_ > 6 ? (int)contact_dof_ - 6 : 0;
        torque_contact_.assign(model_dof_, 0.0);
        if (k > 0 && !ts_.empty())
            for (unsigned i = 0; i < model_dof_; i++)
                for (int j = 0; j < k; j++) torque_contact_[i] += NwJw(i, j) * ts_.back().contact_qp_[j];
        redistributed_ = false;
        return diag_[1];
    }
    int CalcContactRedistribute(bool hqp = true, bool init = true) {
        if (hqp != hqp_) { hqp_ = hqp; dirty_ = true; }  // hqp = false: closed-form ContactRedistributetwomod (dwbc.cpp:1570-1619)
        if (!refresh(init)) return 0;
        torque_contact_ = tau_contact_final_;
        redistributed_ = true;
        return diag_[2];
    }
    // ---- reduced (centroidal) dynamics model: the Reduced* call sequence (dwbc.h:411-416,
    //      tests/sp_test/redu_dyn_test.cpp:263-298).  One fused launch serves the whole sequence, like the full model.
    void ReducedDynamicsCalculate(bool = false) { if (!reduced_) dirty_ = true; reduced_ = true; }
    int ReducedCalcContactConstraint() { reduced_on(); return refresh() ? diag_[0] : 0; }
    void ReducedCalcGravCompensation() { reduced_on(); refresh(); }
    void ReducedCalcTaskSpace(bool = true) { reduced_on(); }
    int ReducedCalcTaskControlTorque(bool hqp = true, bool init = true, bool = true) {
        if (!hqp) { std::cout << "libdwbc_amd : hqp=false is not on the device path" << std::endl; return 0; }
        reduced_on();
        if (!refresh(init)) return 0;
        torque_contact_.assign(model_dof_, 0.0);  // the reduced cascade does not touch torque_contact_ (dwbc.cpp:3255-3446)
        return diag_[1];
    }
    int ReducedCalcContactRedistribute(bool hqp = true, bool init = true) {
        if (!hqp) { std::cout << "libdwbc_amd : hqp=false is not on the device path" << std::endl; return 0; }
        reduced_on();
        if (!refresh(init)) return 0;
        if (contact_dof_ <= 6) return 0;  // dwbc.cpp:3760-3769: nothing happens in single support
        torque_contact_ = tau_contact_final_;
        return diag_[2];
    }
    // back to the full model for the next CalcContactConstraint / CalcTaskControlTorque
    void UseFullDynamics() { if (reduced_) dirty_ = true; reduced_ = false; }

    Vec3 getZMP(const Vec &contact_force) {  // dwbc.h:304, src/dwbc.cpp:898-939 (packed wrench of the active contacts)
        double tot = 0.0, z[3] = {0, 0, 0};
        unsigned a = 0;
        for (auto &c : cc_) if (c.contact) { tot += contact_force[6 * a + 2]; a++; }
        a = 0;
        for (auto &c : cc_) {
            if (!c.contact) continue;
            const double fz = contact_force[6 * a + 2];
            c.zmp_pos = c.xc_pos;
            if (!(fz > -1.0e-3)) { c.zmp_pos.v[0] += -contact_force[6 * a + 4] / fz; c.zmp_pos.v[1] += contact_force[6 * a + 3] / fz; }
            for (int x = 0; x < 3; x++) z[x] += c.zmp_pos.v[x] * fz / tot;
            a++;
        }
        return Vec3(z[0], z[1], z[2]);
    }
    // ---- LQP formulation on the generic HQP class (dwbc.h:365-371; src/dwbc.cpp:4304-4452): y = [qddot; f_c]
    int ConfigureLQP(HQP &hqp, bool = true) {
        if (!refresh()) return 0;
        if (!hqp.handle() || hqp.acceleration_size_ != (int)system_dof_ || hqp.contact_size_ != (int)contact_dof_) hqp.initialize(system_dof_, 0, contact_dof_, device_);
        if (!dwbc_batch_configure_lqp(batch_, hqp.handle())) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        hqp.fetch();
        return 1;
    }
    int CalcControlTorqueLQP(HQP &hqp, bool init = true) {
        hqp.solveSequential(init);
        for (auto &hh : hqp.hqp_hs_) if (!hh.qp_status_) return 0;
        return 1;
    }
    // torque of the LQP answer, as the reference's harness forms it (tests/sp_test/jacc_compare.cpp:416-418)
    Vec LQPTorque(HQP &hqp) {  // model_dof_ torques; after ConfigureLQP_R: reduced_model_dof_ (chain torques | centroidal wrench)
        Vec tau(hqp.acceleration_size_ - 6, 0.0);
        if (!dwbc_batch_lqp_torque(batch_, hqp.handle(), tau.data())) std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl;
        return tau;
    }
    // CalcSingleTaskTorqueWithJACC_QP(ts_[level], init) (src/dwbc.cpp:3772-3945); levels in order
    int CalcSingleTaskTorqueWithJACC_QP(int level, bool = true) {
        if (!refresh()) return 0;
        if (!jacc_h_.handle() || jacc_h_.acceleration_size_ != (int)system_dof_ || jacc_h_.contact_size_ != (int)contact_dof_) jacc_h_.initialize(system_dof_, 0, contact_dof_, device_);
        if (!dwbc_batch_solve_jacc(batch_, jacc_h_.handle(), level)) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        return fetch_jacc(level, system_dof_);
    }
    // ---- the same formulations on the reduced model (dwbc.h; src/dwbc.cpp:4455-4760): after ReducedDynamicsCalculate().
    //      hqp: y = [qddot_R (reduced_system_dof_); f_c];  hqp_nc: y = accelerations of the nc_dof non-contact joints
    int ConfigureLQP_R(HQP &hqp, bool = true) {
        reduced_on();
        if (!refresh()) return 0;
        if (!hqp.handle() || hqp.acceleration_size_ != (int)reduced_system_dof_ || hqp.contact_size_ != (int)contact_dof_) hqp.initialize(reduced_system_dof_, 0, contact_dof_, device_);
        if (!dwbc_batch_configure_lqp_r(batch_, hqp.handle())) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        hqp.fetch();
        return 1;
    }
    int CalcControlTorqueLQP_R(HQP &hqp, bool init = true) { return CalcControlTorqueLQP(hqp, init); }
    // q_acc of the reference (= hqp_r.hqp_hs_.back().y_ans_.head(acceleration_size_)) is read on the device from hqp_r itself;
    // nc_level: the 6-D task level on a non-contact link (the reference reads ts_[1])
    int ConfigureLQP_R_NC(HQP &hqp_nc, HQP &hqp_r, int nc_level = 1, bool = true) {
        if (!hqp_nc.handle() || hqp_nc.acceleration_size_ != (int)nc_dof) hqp_nc.initialize(nc_dof, 0, 0, device_);
        if (!dwbc_batch_configure_lqp_r_nc(batch_, hqp_nc.handle(), hqp_r.handle(), nc_level)) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        hqp_nc.fetch();
        return 1;
    }
    int CalcControlTorqueLQP_R_NC(HQP &hqp_nc, bool init = true) {
        hqp_nc.solvefirst(init);
        hqp_nc.solveSequential(init);
        for (auto &hh : hqp_nc.hqp_hs_) if (!hh.qp_status_) return 0;
        return 1;
    }
    // CalcSingleTaskTorqueWithJACC_QP_R(ts_[level]) / _R_NC(ts_[level], acc_qp_ of the reduced level src_level): results in
    // ts_[level].acc_qp_ / torque_qp_ / contact_qp_ (gacc_qp_ for _R_NC) / f_star_qp_
    int CalcSingleTaskTorqueWithJACC_QP_R(int level, bool = true) {
        reduced_on();
        if (!refresh()) return 0;
        if (!jacc_h_.handle() || jacc_h_.acceleration_size_ != (int)reduced_system_dof_ || jacc_h_.contact_size_ != (int)contact_dof_) jacc_h_.initialize(reduced_system_dof_, 0, contact_dof_, device_);
        if (!dwbc_batch_solve_jacc_r(batch_, jacc_h_.handle(), level)) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        return fetch_jacc(level, reduced_system_dof_);
    }
    int CalcSingleTaskTorqueWithJACC_QP_R_NC(int level, int src_level, bool = true) {
        if (!jacc_nc_h_.handle() || jacc_nc_h_.acceleration_size_ != (int)nc_dof) jacc_nc_h_.initialize(nc_dof, 0, 0, device_);
        if (!dwbc_batch_solve_jacc_r_nc(batch_, jacc_nc_h_.handle(), level, src_level)) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        TaskSpaceView &t = ts_[level];
        t.acc_qp_.assign(nc_dof, 0.0); t.torque_qp_.assign(nc_dof, 0.0); t.gacc_qp_.assign(6, 0.0); t.f_star_qp_.assign(6, 0.0);
        int st = 0;
        dwbc_batch_get_jacc_nc(batch_, DWBC_JACC_ACC, t.acc_qp_.data(), nc_dof * 8);
        dwbc_batch_get_jacc_nc(batch_, DWBC_JACC_TORQUE, t.torque_qp_.data(), nc_dof * 8);
        dwbc_batch_get_jacc_nc(batch_, DWBC_JACC_CONTACT, t.gacc_qp_.data(), 6 * 8);
        dwbc_batch_get_jacc_nc(batch_, DWBC_JACC_FSTAR_QP, t.f_star_qp_.data(), 6 * 8);
        dwbc_batch_get_jacc_nc(batch_, DWBC_JACC_STATUS, &st, sizeof(int));
        return st;
    }
    int CalcAll(bool init = true) { int ok = refresh(init); torque_contact_ = tau_contact_final_; return ok && diag_[0] && diag_[1] && diag_[2]; }
    template <class V, class = decltype(std::declval<const V &>().data()), class = decltype(std::declval<const V &>().rows())>
    Vec getContactForce(const V &command_torque) { return getContactForce(dwbc_amd::as_vec(command_torque)); }
    Vec getContactForce(const Vec &command_torque) {  // wbd.cpp:268-271: J_C_INV_T[:,6:] tau - P_C
        if (general_) {  // the device's own getContactForce(torque_grav_ + torque_task_ + torque_contact_): J_C_INV_T is not mirrored here
            refresh();
            (void)command_torque;
            return Vec(wrench_.begin(), wrench_.begin() + (contact_dof_ <= wrench_.size() ? contact_dof_ : wrench_.size()));
        }
        Vec f(contact_dof_, 0.0);
        for (unsigned c = 0; c < contact_dof_; c++) {
            double s = -P_C[c];
            for (unsigned j = 0; j < model_dof_; j++) s += J_C_INV_T(c, 6 + j) * command_torque[j];
            f[c] = s;
        }
        return f;
    }

  private:
    dwbc_model *model_ = nullptr;
    dwbc_batch *batch_ = nullptr;
    std::vector<uint8_t> flags_;
    bool dirty_ = true, redistributed_ = false, reduced_ = false, hqp_ = true, contact_ok_ = true;
    // general_: more than two contacts flagged at some point, or a task level wider than six dof -- every solve then runs the
    // general-contact kernel (libdwbc_amd/csrc/dwbc_cycle_gc.h), which keeps no dump record: torque_grav_ / torque_task_ / torque_contact_,
    // getContactForce(total torque) and the int returns are served; the public matrices (A_, J_C, Lambda_contact, ts_[i].J_kt_ ...) are not
    bool general_ = false;
    Vec wrench_;
    void enter_general() { if (!general_) { general_ = true; dwbc_batch_enable_dump(batch_, 0); dirty_ = true; } }
    HQP jacc_h_, jacc_nc_h_;  // solver objects of the JACC entry points
    int fetch_jacc(int level, int n) {
        TaskSpaceView &t = ts_[level];
        t.acc_qp_.assign(n, 0.0); t.torque_qp_.assign(n - 6, 0.0); t.contact_qp_.assign(12, 0.0); t.f_star_qp_.assign(6, 0.0);
        int st = 0;
        dwbc_batch_get_jacc(batch_, level, DWBC_JACC_ACC, t.acc_qp_.data(), n * 8);
        dwbc_batch_get_jacc(batch_, level, DWBC_JACC_TORQUE, t.torque_qp_.data(), (n - 6) * 8);
        dwbc_batch_get_jacc(batch_, level, DWBC_JACC_CONTACT, t.contact_qp_.data(), 12 * 8);
        dwbc_batch_get_jacc(batch_, level, DWBC_JACC_FSTAR_QP, t.f_star_qp_.data(), 6 * 8);
        dwbc_batch_get_jacc(batch_, level, DWBC_JACC_STATUS, &st, sizeof(int));
        t.contact_qp_.resize(contact_dof_); t.f_star_qp_.resize(t.task_dof_);
        return st;
    }
    double sent_time_ = -1.0e300;
    void reduced_on() { if (!reduced_) { reduced_ = true; dirty_ = true; } }
    int diag_[96] = {0};
    Vec tau_contact_final_;

    int device_ = 0;
    void release() {
        if (batch_) dwbc_batch_destroy(batch_);
        if (model_) dwbc_model_destroy(model_);
        batch_ = nullptr; model_ = nullptr;
    }
    void replace_model(dwbc_model *edited, bool verbose) {
        if (!edited) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return; }
        if (!cc_.empty() || !ts_.empty()) std::cout << "libdwbc_amd : model surgery after AddContactConstraint / AddTaskSpace: they are cleared (link ids moved)" << std::endl;
        cc_.clear(); ts_.clear();
        const bool fl = is_floating_;
        release();
        dirty_ = true; contact_ok_ = true;
        adopt_model(edited, fl, verbose ? 1 : 0);
    }
    Mat fetch(int field, int r, int c, int src_cols = -1) {
        size_t nb = dwbc_batch_field_bytes(batch_, field);
        std::vector<double> buf(nb / 8);
        Mat m(r, c);
        if (!dwbc_batch_get(batch_, field, buf.data(), nb)) return m;
        if (src_cols < 0) src_cols = c;
        for (int i = 0; i < r; i++) for (int j = 0; j < c; j++) m(i, j) = buf[(size_t)i * src_cols + j];
        return m;
    }
    int refresh(bool init = true) {
        if (!batch_ || !contact_ok_) return 0;  // a refused SetContact leaves nothing to solve
        for (size_t l = 0; l < ts_.size(); l++)  // task-link trajectories / gains set since the last launch
            for (size_t j = 0; j < ts_[l].task_link_.size(); j++) {
                TaskLinkView &tl = ts_[l].task_link_[j];
                if (!tl.dirty) continue;
                dwbc_batch_set_task_gain(batch_, (int)l, (int)j, tl.gain, tl.gain + 3, tl.gain + 6, tl.gain + 9, tl.gain + 12, tl.gain + 15);
                if (tl.traj_pos_set || tl.traj_rot_set) dwbc_batch_set_trajectory(batch_, (int)l, (int)j, tl.rec);
                tl.dirty = false;
                dirty_ = true;
            }
        if (control_time_ != sent_time_) { dwbc_batch_set_control_time(batch_, &control_time_); sent_time_ = control_time_; dirty_ = true; }
        if (!dirty_) return 1;
        if (!dwbc_batch_solve(batch_, (hqp_ ? DWBC_SOLVE_HQP : 0) | (init ? DWBC_SOLVE_INIT : 0) | (reduced_ ? DWBC_SOLVE_REDUCED : 0))) { std::cout << "libdwbc_amd : " << dwbc_last_error() << std::endl; return 0; }
        const int n = system_dof_, m = model_dof_, cd = contact_dof_ > 12 ? 12 : (int)contact_dof_, k = cd > 6 ? cd - 6 : 0;
        std::vector<double> tau(3 * m);
        dwbc_batch_get(batch_, DWBC_TAU, tau.data(), tau.size() * 8);
        torque_grav_.assign(tau.begin(), tau.begin() + m);
        torque_task_.assign(tau.begin() + m, tau.begin() + 2 * m);
        tau_contact_final_.assign(tau.begin() + 2 * m, tau.end());
        dwbc_batch_get(batch_, DWBC_DIAG, diag_, sizeof(int) * 90);
        if (general_) {
            wrench_.assign(dwbc_batch_field_bytes(batch_, DWBC_WRENCH) / 8, 0.0);
            dwbc_batch_get(batch_, DWBC_WRENCH, wrench_.data(), wrench_.size() * 8);
            dirty_ = false;
            return 1;
        }
        A_ = fetch(DWBC_A, n, n); A_inv_ = fetch(DWBC_A_INV, n, n); A_inv_N_C = fetch(DWBC_A_INV_N_C, n, n);
        J_C = fetch(DWBC_J_C, cd, n); J_C_INV_T = fetch(DWBC_J_C_INV_T, cd, n);
        Lambda_contact = fetch(DWBC_LAMBDA_C, cd, cd, cd);
        if (!reduced_) { W_inv = fetch(DWBC_W_INV, m, m); NwJw = fetch(DWBC_NWJW, m, k, k > 0 ? k : 1); }
        Mat g = fetch(DWBC_G, 1, n), pc = fetch(DWBC_P_C, 1, cd, 12);
        G_ = g.d; P_C = pc.d;
        CMM_ = fetch(DWBC_CMM, 6, n); J_com_ = fetch(DWBC_J_COM, 6, n); com_inertia_ = fetch(DWBC_COM_INERTIA, 3, 3);
        { Mat cp = fetch(DWBC_COM, 1, 3); com_pos = Vec3(cp.d[0], cp.d[1], cp.d[2]); }
        {
            Mat xp = fetch(DWBC_CONTACT_POS, 2, 3), xr = fetch(DWBC_CONTACT_ROT, 2, 9);
            int a = 0;
            for (auto &c : cc_) {
                if (!c.contact || a >= 2) continue;
                c.xc_pos = Vec3(xp(a, 0), xp(a, 1), xp(a, 2));
                c.rotm = Mat(3, 3);
                for (int x = 0; x < 9; x++) c.rotm.d[x] = xr(a, x);
                a++;
            }
        }
        {
            Mat bb = fetch(DWBC_B, 1, n), lr = fetch(DWBC_LINK_R, 48, 9), lp = fetch(DWBC_LINK_P, 48, 3), lv = fetch(DWBC_LINK_V, 48, 3), lw = fetch(DWBC_LINK_W, 48, 3);
            B_ = bb.d;
            link_.resize(link_num_);
            for (unsigned i = 0; i < link_num_; i++) {
                link_[i].xpos = Vec3(lp(i, 0), lp(i, 1), lp(i, 2));
                link_[i].v = Vec3(lv(i, 0), lv(i, 1), lv(i, 2));
                link_[i].w = Vec3(lw(i, 0), lw(i, 1), lw(i, 2));
                link_[i].rotm = Mat(3, 3);
                for (int a = 0; a < 9; a++) link_[i].rotm.d[a] = lr(i, a);
            }
        }
        W = Mat(m, m);
        for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) W(i, j) = A_inv_N_C(6 + i, 6 + j);
        N_C = Mat(n, n);  // N_C = I - J_C^T J_C_INV_T (wbd.cpp:117)
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int c = 0; c < cd; c++) s -= J_C(c, i) * J_C_INV_T(c, j);
            N_C(i, j) = s;
        }
        if (reduced_) {  // W_inv, NwJw, ts_[i].J_kt_ ... are full-model fields; the reduced model's own:
            int vc = 0, ncd = 0;
            if (dwbc_batch_reduced_dims(batch_, &vc, &ncd)) {
                vc_dof = vc; nc_dof = ncd; co_dof = vc - 6; reduced_model_dof_ = vc; reduced_system_dof_ = vc + 6;
                const int rs = vc + 6;
                A_R = fetch(DWBC_A_R, rs, rs, 24); A_R_inv = fetch(DWBC_A_R_INV, rs, rs, 24);
                { Mat gr = fetch(DWBC_G_R, 1, rs, 24); G_R = gr.d; }
                J_I_nc_ = fetch(DWBC_J_I_NC, 6, ncd, n - 12); J_I_nc_inv_T = fetch(DWBC_J_I_NC_INV_T, 6, ncd, n - 12);
                const Mat jt = fetch(DWBC_J_TASK, 4 * 6, n);
                for (size_t l = 0; l < ts_.size(); l++) {
                    ts_[l].J_task_ = Mat(ts_[l].task_dof_, n);
                    for (int r = 0; r < ts_[l].task_dof_; r++)
                        for (int c = 0; c < n; c++) ts_[l].J_task_(r, c) = jt((int)l * 6 + r, c);
                }
            }
            dirty_ = false;
            return 1;
        }
        Mat fq = fetch(DWBC_FSTAR_QP, 4, 6), cq = fetch(DWBC_CONTACT_QP, 4, 6), cr = fetch(DWBC_CF_REDIS, 1, 6);
        cf_redis_qp_.assign(cr.d.begin(), cr.d.begin() + k);
        std::vector<double> jt(dwbc_batch_field_bytes(batch_, DWBC_J_TASK) / 8), lt(dwbc_batch_field_bytes(batch_, DWBC_LAMBDA_TASK) / 8), jk(dwbc_batch_field_bytes(batch_, DWBC_J_KT) / 8);
        dwbc_batch_get(batch_, DWBC_J_TASK, jt.data(), jt.size() * 8);
        dwbc_batch_get(batch_, DWBC_LAMBDA_TASK, lt.data(), lt.size() * 8);
        dwbc_batch_get(batch_, DWBC_J_KT, jk.data(), jk.size() * 8);
        for (size_t l = 0; l < ts_.size(); l++) {
            const int t = ts_[l].task_dof_;
            ts_[l].f_star_qp_.assign(&fq.d[l * 6], &fq.d[l * 6] + t);
            ts_[l].contact_qp_.assign(&cq.d[l * 6], &cq.d[l * 6] + k);
            ts_[l].J_task_ = Mat(t, n); ts_[l].Lambda_task_ = Mat(t, t); ts_[l].J_kt_ = Mat(m, t);
            for (int i = 0; i < t; i++) for (int j = 0; j < n; j++) ts_[l].J_task_(i, j) = jt[l * 6 * n + i * n + j];
            for (int i = 0; i < t; i++) for (int j = 0; j < t; j++) ts_[l].Lambda_task_(i, j) = lt[l * 36 + i * t + j];
            for (int i = 0; i < m; i++) for (int j = 0; j < t; j++) ts_[l].J_kt_(i, j) = jk[l * m * 6 + i * t + j];
            ts_[l].qp_error = (diag_[1] == 0 && diag_[3] == (int)l) ? 1 : 0;
        }
        dirty_ = false;
        return 1;
    }
};

}  // namespace DWBC
