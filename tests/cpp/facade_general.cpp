// The call sequences of reference tests/sp_test/data_confirmation.cpp:60-107 (four levels with both hands on the last one: two
// AddTaskSpace(3, TASK_LINK_6D, ...) calls and a 12-vector f*; a cold solve, then CalcTaskControlTorque(hqp, false, false) /
// CalcContactRedistribute(hqp, false) warm) and of a third simultaneous contact (SetContact(c1, c2, c3), :91) written against the
// drop-in facade include/dwbc_amd.hpp.  Prints the torques and getContactForce as JSON for tests/test_facade_cpp.py.
#include <cstdio>
#include <cmath>
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
    if (argc < 2) { fprintf(stderr, "usage: facade_general <urdf>\n"); return 2; }
    const double q2[40] = {-0.0325, -0.0579, 0.7273, 0.0194, -0.0118, -0.0008, -0.0006, 0.0698, -0.7835, 1.6487, -0.8420, -0.0911,
                           -0.0007, 0.0767, -0.7963, 1.6742, -0.8549, -0.1150, -0.0001, -0.0003, 0.0204,
                           0.2998, 0.3001, 1.5000, -1.2701, -1.0507, 0.0000, -1.0000, 0.0000, -0.0000, 0.0003,
                           -0.2998, -0.3060, -1.5001, 1.2700, 1.0848, 0.0000, 1.0000, 0.0000, 0.9997};
    Vec q(q2, q2 + 40), qdot(39, 0.0), qddot(39, 0.0);
    {  // (the harness's four-digit quaternion is not of unit length: normalised here, as tests/test_facade_cpp.py does for the restatement)
        const double nq = std::sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5] + q[39] * q[39]);
        q[3] /= nq; q[4] /= nq; q[5] /= nq; q[39] /= nq;
    }
    printf("{\n");
    {
        RobotData rd2_;
        rd2_.LoadModelData(argv[1], true, false);
        if (rd2_.system_dof_ != 39) { fprintf(stderr, "model load failed\n"); return 3; }
        rd2_.UpdateKinematics(q, qdot, qddot);
        rd2_.AddContactConstraint(6, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.12, 0.06);
        rd2_.AddContactConstraint(12, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.12, 0.06);
        rd2_.AddContactConstraint(23, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.04, 0.04);
        rd2_.AddContactConstraint(31, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.04, 0.04);
        rd2_.AddTaskSpace(0, TASK_LINK_POSITION, 0, Vec3());
        rd2_.AddTaskSpace(1, TASK_LINK_ROTATION, "upperbody_link", Vec3());
        rd2_.AddTaskSpace(2, TASK_LINK_ROTATION, 25, Vec3());
        rd2_.AddTaskSpace(3, TASK_LINK_6D, "L_Wrist2_Link", Vec3());
        rd2_.AddTaskSpace(3, TASK_LINK_6D, "R_Wrist2_Link", Vec3());
        rd2_.SetTorqueLimit(Vec(rd2_.model_dof_, 300.0));
        rd2_.SetContact(true, true, false, false);
        int ok_c = rd2_.CalcContactConstraint();
        rd2_.SetTaskSpace(0, Vec{0.3142, -1.8202, -1.7750});
        rd2_.SetTaskSpace(1, Vec{-1.78677, 0.84977, 0.10850});
        rd2_.SetTaskSpace(2, Vec{-0.85340, 0.85992, 0.12655});
        rd2_.SetTaskSpace(3, Vec{0.40251, 0.39975, 0.75672, -0.82841, 3.03652, 0.08954, 0.27585, 0.37898, 0.93234, -0.95724, 4.38036, 0.25202});
        rd2_.CalcGravCompensation();
        int ok_t = rd2_.CalcTaskControlTorque(true, true);
        int ok_r = rd2_.CalcContactRedistribute(true, true);
        printf("\"dims\": [%d, %d, %d],\n", (int)rd2_.ts_.size(), rd2_.ts_[3].task_dof_, (int)rd2_.contact_dof_);
        printf("\"ok\": [%d, %d, %d],\n", ok_c, ok_t, ok_r);
        print_vec("torque_grav_", rd2_.torque_grav_);
        print_vec("torque_task_", rd2_.torque_task_);
        print_vec("torque_contact_", rd2_.torque_contact_);
        Vec total(rd2_.model_dof_);
        for (unsigned i = 0; i < rd2_.model_dof_; i++) total[i] = rd2_.torque_grav_[i] + rd2_.torque_task_[i] + rd2_.torque_contact_[i];
        print_vec("contact_force", rd2_.getContactForce(total));
        // the warm repetitions of the harness (init = false): the same point
        int warm_ok = 0;
        for (int i = 0; i < 3; i++) {
            rd2_.SetContact(true, true, false);
            rd2_.CalcContactConstraint();
            rd2_.CalcGravCompensation();
            warm_ok += rd2_.CalcTaskControlTorque(true, false, false);
            warm_ok += rd2_.CalcContactRedistribute(true, false);
        }
        printf("\"warm_ok\": %d,\n", warm_ok);
        print_vec("warm_torque_task_", rd2_.torque_task_);
    }
    {
        // a third simultaneous contact through the drop-in class: feet + left hand, pelvis 6D and upper-body rotation
        RobotData rd_;
        rd_.LoadModelData(argv[1], true, false);
        rd_.UpdateKinematics(q, qdot, qddot);
        rd_.AddContactConstraint(6, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.12, 0.06);
        rd_.AddContactConstraint(12, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.12, 0.06);
        rd_.AddContactConstraint(23, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.04, 0.04);
        rd_.AddContactConstraint(31, CONTACT_6D, Vec3(0.03, 0, -0.1585), Vec3(0, 0, 1), 0.04, 0.04);
        rd_.AddTaskSpace(0, TASK_LINK_6D, 0, Vec3());
        rd_.AddTaskSpace(1, TASK_LINK_ROTATION, "upperbody_link", Vec3());
        rd_.SetTorqueLimit(Vec(rd_.model_dof_, 300.0));
        rd_.SetContact(true, true, true);
        int ok_c = rd_.CalcContactConstraint();
        rd_.SetTaskSpace(0, Vec{0.1, 0.4, 0.1, 0.1, -0.1, 0.1});
        rd_.SetTaskSpace(1, Vec{0.1, -0.1, 0.1});
        rd_.CalcGravCompensation();
        int ok_t = rd_.CalcTaskControlTorque(true, true);
        int ok_r = rd_.CalcContactRedistribute(true, true);
        printf("\"c3_ok\": [%d, %d, %d, %d],\n", ok_c, ok_t, ok_r, (int)rd_.contact_dof_);
        print_vec("c3_torque_grav_", rd_.torque_grav_);
        print_vec("c3_torque_task_", rd_.torque_task_);
        print_vec("c3_torque_contact_", rd_.torque_contact_);
        Vec total(rd_.model_dof_);
        for (unsigned i = 0; i < rd_.model_dof_; i++) total[i] = rd_.torque_grav_[i] + rd_.torque_task_[i] + rd_.torque_contact_[i];
        print_vec("c3_contact_force", rd_.getContactForce(total), true);
    }
    printf("}\n");
    return 0;
}
