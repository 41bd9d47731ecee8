// Mirror of reference tests/dwbc_test.cpp:29-131 (CASE 1 / CASE 2) written against the drop-in facade
// include/dwbc_amd.hpp.  Prints torque_grav_, torque_task_, torque_contact_ (after CalcTaskControlTorque and after
// CalcContactRedistribute) and a few matrices' checksums as JSON for tests/test_facade_cpp.py.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "dwbc_amd.hpp"

using namespace DWBC;

static void print_vec(const char *name, const Vec &v, bool last = false) {
    printf("\"%s\": [", name);
    for (size_t i = 0; i < v.size(); i++) printf("%s%.17g", i ? ", " : "", v[i]);
    printf("]%s\n", last ? "" : ",");
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: facade_case <urdf> <case 1|2>\n"); return 2; }
    const int cs = atoi(argv[2]);
    RobotData rd_;
    rd_.LoadModelData(argv[1], true, false);
    if (rd_.system_dof_ != 39) { fprintf(stderr, "model load failed\n"); return 3; }
    if (argc > 3 && std::string(argv[3]) == "api") rd_.printLinkInfo();  // (before the JSON block)
    Vec q(rd_.system_dof_ + 1, 0.0), qdot(rd_.system_dof_, 0.0), qddot(rd_.system_dof_, 0.0);
    const double q1[40] = {0, 0, 0.92983, 0, 0, 0, 0.0, 0.0, -0.24, 0.6, -0.36, 0.0, 0.0, 0.0, -0.24, 0.6, -0.36, 0.0, 0, 0, 0,
                           0.3, 0.3, 1.5, -1.27, -1, 0, -1, 0, 0, 0, -0.3, -0.3, -1.5, 1.27, 1, 0, 1, 0, 1};
    const double q2[40] = {0, 0, 0.92983, 0, 0, 0, 0.1, 0.0, -0.24, 0.5, -0.6, 0.0, 0.05, 0.0, -0.21, 0.7, -0.31, 0.0, 0, 0, 0,
                           0.2, 0.5, 1.5, -1.27, -1.2, 0, -1, 0, 0, 0, -0.3, -0.3, -1.5, 1.27, 1.3, 0.1, 1.3, 0, 1};
    for (int i = 0; i < 40; i++) q[i] = cs == 1 ? q1[i] : q2[i];
    Vec fstar_1 = cs == 1 ? Vec{0.1, 4.0, 0.1, 0.1, -0.1, 0.1} : Vec{0.4, 2.0, 0.1, 0.3, -0.1, 0.1};
    Vec fstar_2 = cs == 1 ? Vec{0.1, -0.1, 0.1} : Vec{0.1, 0.1, 0.1};

    rd_.UpdateKinematics(q, qdot, qddot);
    int left_foot_id = 6, right_foot_id = 12;
    rd_.AddContactConstraint(left_foot_id, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.15, 0.075);
    rd_.AddContactConstraint(right_foot_id, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.15, 0.075);
    rd_.AddContactConstraint(23, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.04, 0.04);
    rd_.AddContactConstraint(31, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.04, 0.04);
    rd_.AddTaskSpace(0, TASK_LINK_6D, 0, Vec3());
    const bool lqp_r = argc > 3 && std::string(argv[3]) == "lqp_r";
    if (lqp_r) {  // the reduced LQP / JACC harness shape: a 6-D task on a non-contact link (tests/sp_test/jacc_compare.cpp:456-487)
        rd_.AddTaskSpace(1, TASK_LINK_6D, "upperbody_link", Vec3());
        fstar_2 = Vec{0.05, -0.1, 0.02, fstar_2[0], fstar_2[1], fstar_2[2]};
    } else
        rd_.AddTaskSpace(1, TASK_LINK_ROTATION, "upperbody_link", Vec3());
    const bool with_reduced = lqp_r || (argc > 3 && std::string(argv[3]) == "reduced");
    if (!with_reduced) rd_.SetTorqueLimit(Vec(rd_.model_dof_, 300.0));  // the reduced sequence runs without it (redu_dyn_test.cpp:63)
    rd_.UpdateKinematics(q, qdot, qddot);
    rd_.SetContact(true, true);
    int ok_c = rd_.CalcContactConstraint();
    rd_.SetTaskSpace(0, fstar_1);
    rd_.SetTaskSpace(1, fstar_2);
    Vec tg = rd_.CalcGravCompensation();
    int ok_t = rd_.CalcTaskControlTorque(true);
    Vec tc_before = rd_.torque_contact_;
    int ok_r = rd_.CalcContactRedistribute(true);
    Vec total(rd_.model_dof_);
    for (unsigned i = 0; i < rd_.model_dof_; i++) total[i] = rd_.torque_grav_[i] + rd_.torque_task_[i] + rd_.torque_contact_[i];
    Vec cf = rd_.getContactForce(total);
    printf("{\n\"ok\": [%d, %d, %d],\n", ok_c, ok_t, ok_r);
    printf("\"dims\": [%u, %u, %u, %u],\n", rd_.system_dof_, rd_.model_dof_, rd_.contact_dof_, rd_.contact_link_num_);
    print_vec("torque_grav_", tg);
    print_vec("torque_task_", rd_.torque_task_);
    print_vec("torque_contact_before_redis", tc_before);
    print_vec("torque_contact_", rd_.torque_contact_);
    print_vec("contact_force", cf);
    print_vec("A_inv_", rd_.A_inv_.d);
    print_vec("N_C", rd_.N_C.d);
    print_vec("W", rd_.W.d);
    print_vec("NwJw", rd_.NwJw.d);
    if (with_reduced) {
        // reference tests/sp_test/redu_dyn_test.cpp:263-298 on the same state
        rd_.SetContact(true, true);
        rd_.ReducedDynamicsCalculate();
        int rk_c = rd_.ReducedCalcContactConstraint();
        rd_.ReducedCalcGravCompensation();
        rd_.ReducedCalcTaskSpace();
        int rk_t = rd_.ReducedCalcTaskControlTorque(true, true, false);
        int rk_r = rd_.ReducedCalcContactRedistribute(true, true);
        printf("\"reduced_ok\": [%d, %d, %d],\n", rk_c, rk_t, rk_r);
        print_vec("reduced_torque_grav_", rd_.torque_grav_);
        print_vec("reduced_torque_task_", rd_.torque_task_);
        print_vec("reduced_torque_contact_", rd_.torque_contact_);
    }
    if (lqp_r) {
        // reference tests/sp_test/jacc_compare.cpp:456-487 and dof_comparison_jacc.cpp:358-362 through the facade
        DWBC::HQP hqp_, hqp_nc_;
        rd_.ReducedDynamicsCalculate();
        int ok_cfg = rd_.ConfigureLQP_R(hqp_);
        int ok_lqp = rd_.CalcControlTorqueLQP_R(hqp_, true);
        int ok_ncc = rd_.ConfigureLQP_R_NC(hqp_nc_, hqp_, 1);
        int ok_nc = rd_.CalcControlTorqueLQP_R_NC(hqp_nc_, true);
        printf("\"lqp_r_ok\": [%d, %d, %d, %d, %d, %d],\n", ok_cfg, ok_lqp, ok_ncc, ok_nc, (int)hqp_.hqp_hs_.size(), (int)hqp_nc_.hqp_hs_.size());
        printf("\"reduced_dims\": [%u, %u, %u, %u, %u],\n", rd_.vc_dof, rd_.nc_dof, rd_.co_dof, rd_.reduced_model_dof_, rd_.reduced_system_dof_);
        print_vec("lqp_r_y", hqp_.hqp_hs_.back().y_ans_);
        print_vec("lqp_r_torque", rd_.LQPTorque(hqp_));
        print_vec("lqp_nc_y", hqp_nc_.hqp_hs_.back().y_ans_);
        int ok_j = rd_.CalcSingleTaskTorqueWithJACC_QP_R(0);
        int ok_jn = rd_.CalcSingleTaskTorqueWithJACC_QP_R_NC(1, 0);
        printf("\"jacc_r_ok\": [%d, %d],\n", ok_j, ok_jn);
        print_vec("jacc_r_acc", rd_.ts_[0].acc_qp_);
        print_vec("jacc_r_torque", rd_.ts_[0].torque_qp_);
        print_vec("jacc_nc_acc", rd_.ts_[1].acc_qp_);
        print_vec("G_R", rd_.G_R);
    }
    if (argc > 3 && std::string(argv[3]) == "lqp") {
        // reference tests/sp_test/jacc_compare.cpp:386-418: ConfigureLQP, CalcControlTorqueLQP, torque of the answer
        DWBC::HQP hqp_;
        int ok_cfg = rd_.ConfigureLQP(hqp_);
        int ok_lqp = rd_.CalcControlTorqueLQP(hqp_, true);
        printf("\"lqp_ok\": [%d, %d, %d],\n", ok_cfg, ok_lqp, (int)hqp_.hqp_hs_.size());
        print_vec("lqp_y", hqp_.hqp_hs_.back().y_ans_);
        print_vec("lqp_torque", rd_.LQPTorque(hqp_));
        printf("\"lqp_null\": [%d, %d, %d, %d],\n", hqp_.hqp_hs_[0].null_space_size_, hqp_.hqp_hs_[1].null_space_size_, hqp_.hqp_hs_[2].null_space_size_, hqp_.hqp_hs_[3].null_space_size_);
    }
    {
        // a caller's own vector type (anything with data() / size(), e.g. Eigen::VectorXd) goes through the same entry points
        struct MyVec { std::vector<double> s; const double *data() const { return s.data(); } size_t size() const { return s.size(); } };
        MyVec f2{fstar_2};
        rd_.SetTaskSpace(1, f2);
        std::vector<float> back = dwbc_amd::to_vector<std::vector<float>>(rd_.torque_grav_);
        printf("\"generic\": [%zu],\n", back.size());
    }
    if (argc > 3 && std::string(argv[3]) == "api") {
        // the rest of RobotData's public interface (include/dwbc.h:259-410)
        print_vec("C_contact", rd_.getContactConstraintMatrix().d);
        print_vec("cam3", rd_.CalcAngularMomentumMatrix().d);
        Mat cmm6;
        rd_.CalcAngularMomentumMatrix(cmm6);
        print_vec("cmm6", cmm6.d);
        // CopyKinematicsData into a second object: it takes over state, contacts and task spaces, then is switched to single support
        RobotData rd2_;
        rd2_.LoadModelData(argv[1], true, false);
        rd_.CopyKinematicsData(rd2_);
        rd2_.SetContact(true, false);
        int ok2 = rd2_.CalcContactConstraint();
        print_vec("rd2_torque_grav_", rd2_.CalcGravCompensation());
        print_vec("rd2_A00", Vec{rd2_.A_(0, 0), rd2_.A_(7, 9), (double)rd2_.contact_dof_, (double)ok2});
        // a custom level and a second link on a level; clearing both sets up a new problem on the same object
        rd_.ClearTaskSpace();
        rd_.ClearContactConstraint();
        rd_.AddContactConstraint(left_foot_id, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.15, 0.075);
        rd_.AddContactConstraint(right_foot_id, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.15, 0.075);
        rd_.AddTaskSpace(0, TASK_LINK_6D, 0, Vec3());
        rd_.AddTaskSpace(1, TASK_LINK_ROTATION, "upperbody_link", Vec3());
        rd_.AddTaskLink(1, TASK_LINK_POSITION, "R_Wrist2_Link", Vec3());
        rd_.SetContact(true, true);
        rd_.SetTaskSpace(0, fstar_1);
        rd_.SetTaskSpace(1, Vec{fstar_2[0], fstar_2[1], fstar_2[2], 0.2, -0.1, 0.3});
        int ok3 = rd_.CalcTaskControlTorque(true);
        printf("\"api_ok\": [%d, %d, %d],\n", ok3, (int)rd_.ts_.size(), rd_.ts_[1].task_dof_);
        print_vec("api_torque_task_", rd_.torque_task_);
        (void)rd_.GetControlTorque();
    }
    print_vec("contact_qp_last", rd_.ts_.back().contact_qp_, true);
    printf("}\n");
    return 0;
}
