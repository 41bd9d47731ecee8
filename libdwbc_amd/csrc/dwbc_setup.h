// dwbc_setup.h -- host-side construction of the shared problem Setup (contacts, task hierarchy, limits).
// Mirrors RobotData::AddContactConstraint / AddTaskSpace / SetTorqueLimit (reference src/dwbc.cpp:377-427,
// 522-601, 254-258).  Used by the C-ABI layer and by the CPU emulation harness under tests/emu.
#pragma once
#include <string>

#include "dwbc_types.h"
#include "dwbc_topo.h"

namespace dwbc {

inline void setup_init(Setup &su, int nb, int ndof, int maxdepth) {
    su = Setup{};
    su.nb = nb;
    su.ndof = ndof;
    su.maxdepth = maxdepth;
    // nWSR of the first attempt + the 10 x nWSR of SolveQPoases' retry leg (src/qp_wrapper.cpp:298-339): the device solver has no
    // "reliable" option set to fall back to, so the two budgets add up
    su.qp_max_iter_task = 1000 + 10000;   // reference src/dwbc.cpp:1080
    su.qp_max_iter_contact = 300 + 3000;  // reference src/dwbc.cpp:1546
    for (int l = 0; l < kMaxLevels; l++)
        for (int j = 0; j < kMaxTaskLinks; j++) su.t_traj_slot[l][j] = -1;
    for (int l = 0; l < kMaxLevels; l++) su.t_custom_slot[l] = -1;
}

// parent body of every body (Model::topo_table()[0..nb)); must be installed before contacts / tasks are added
inline void setup_set_parents(Setup &su, const int *parent) {
    for (int i = 0; i < su.nb && i < kMaxBodies; i++) su.parent[i] = parent[i];
    su.topo_kind = TopoTocabi::matches(su.nb, parent) ? 1 : 0;
}

// dofs on the path from `link` to the floating base: bits 0..5 (base) and bit (b + 5) for every moving body b on the path
inline unsigned long long link_dofmask(const Setup &su, int link) {
    unsigned long long m = 0x3full;
    for (int b = link, guard = 0; b > 0 && guard < kMaxBodies; b = su.parent[b], guard++) m |= 1ull << (b + 5);
    return m;
}

inline int task_mode_dof(int mode) { return mode <= TASK_LINK_6D_CUSTOM_FRAME ? 6 : 3; }  // reference src/task.cpp:14-31

inline void setup_fstar_layout(Setup &su) {
    int off = 0;
    for (int l = 0; l < su.n_levels; l++) {
        int t = 0;
        for (int j = 0; j < su.t_nlinks[l]; j++) t += task_mode_dof(su.t_mode[l][j]);
        if (su.t_custom_slot[l] >= 0) t = su.t_dof[l];  // TASK_CUSTOM: the dof given to AddTaskSpace(h, TASK_CUSTOM, dof)
        su.t_dof[l] = t;
        su.fstar_off[l] = off;
        off += t;
    }
    su.fstar_total = off;
}

// returns contact index or -1 (err filled)
inline int setup_add_contact(Setup &su, int link, int contact_type, const double point[3], double lx, double ly, double mu,
                             double mu_z, std::string &err) {
    if (contact_type != 0) { err = "only CONTACT_6D is implemented on the device path"; return -1; }
    if (link < 0 || link >= su.nb) { err = "bad link id"; return -1; }
    if (su.n_contacts >= kMaxContacts) { err = "too many contacts"; return -1; }
    for (int i = 0; i < su.n_contacts; i++)
        if (su.c_link[i] == link) { err = "Contact Constraint Already Exist for Link"; return -1; }  // src/dwbc.cpp:379-386
    const int i = su.n_contacts++;
    su.c_link[i] = link;
    for (int a = 0; a < 3; a++) su.c_point[i][a] = point[a];
    su.c_lx[i] = lx;
    su.c_ly[i] = ly;
    su.c_mu[i] = mu;
    su.c_muz[i] = mu_z;
    su.c_dofmask[i] = link_dofmask(su, link);
    return i;
}

// AddTaskSpace(heirarchy, TASK_CUSTOM, task_dof) (reference src/dwbc.cpp:522-530): a level whose Jacobian the caller supplies
inline bool setup_add_custom_task(Setup &su, int level, int task_dof, std::string &err) {
    if (level != su.n_levels || level >= kMaxLevels) { err = "bad task level (levels must be added in order)"; return false; }
    if (task_dof < 1 || task_dof > kMaxTaskDof) { err = "custom task dof must be 1..6 on the device path"; return false; }
    su.n_levels++;
    su.t_nlinks[level] = 0;
    su.t_custom_slot[level] = su.n_custom++;
    su.t_dof[level] = task_dof;
    su.t_dofmask[level] = (1ull << su.ndof) - 1ull;
    setup_fstar_layout(su);
    return true;
}

inline bool setup_add_task(Setup &su, int level, int mode, int link, const double *point, std::string &err) {
    if (level >= 0 && level < su.n_levels && su.t_custom_slot[level] >= 0) { err = "cannot add a link to a TASK_CUSTOM level"; return false; }
    if (level < 0 || level > su.n_levels || level >= kMaxLevels) { err = "bad task level (levels must be added in order)"; return false; }
    if (mode < 0 || mode > TASK_LINK_ROTATION_CUSTOM_FRAME) { err = "bad task mode"; return false; }
    if (link < 0 || link > su.nb) { err = "bad link id"; return false; }
    // link == nb: the synthetic "COM" link (reference src/dwbc.cpp:230-231); its Jacobian is jac_com_ = SI_body^-1 CMM_.  The
    // *_CUSTOM_FRAME position modes ask RBDL for a point Jacobian of a body the COM link does not have (dwbc.cpp:739,766)
    if (link == su.nb && (mode == TASK_LINK_6D_CUSTOM_FRAME || mode == TASK_LINK_POSITION_CUSTOM_FRAME)) { err = "custom-frame task on the COM link is undefined in the reference"; return false; }
    for (int l = 0; l < su.n_levels; l++)
        for (int j = 0; j < su.t_nlinks[l]; j++)
            if (su.t_link[l][j] == link) { err = "Task Space Already Exist for Link"; return false; }  // src/dwbc.cpp:536-546
    const int j = level == su.n_levels ? 0 : su.t_nlinks[level];
    if (j >= kMaxTaskLinks) { err = "too many links in one task level"; return false; }
    int cur = 0;
    for (int a = 0; a < j; a++) cur += task_mode_dof(su.t_mode[level][a]);
    // more than 6 dof on a level (two 6D links: src/dwbc.cpp:592-600) routes the batch through the general-contact kernel
    if (cur + task_mode_dof(mode) > kMaxTaskDofWide) { err = "task level exceeds 12 dof"; return false; }
    if (level == su.n_levels) {
        su.n_levels++;
        su.t_nlinks[level] = 0;
    }
    su.t_mode[level][j] = mode;
    su.t_link[level][j] = link;
    for (int a = 0; a < 3; a++) su.t_point[level][j][a] = point ? point[a] : 0.0;
    su.t_nlinks[level]++;
    su.t_dofmask[level] = 0;
    for (int a = 0; a < su.t_nlinks[level]; a++)
        su.t_dofmask[level] |= su.t_link[level][a] == su.nb ? ((1ull << su.ndof) - 1ull) : link_dofmask(su, su.t_link[level][a]);
    su.has_com_task = 0;
    for (int l = 0; l < su.n_levels; l++)
        for (int a = 0; a < su.t_nlinks[l]; a++) su.has_com_task |= su.t_link[l][a] == su.nb;
    setup_fstar_layout(su);
    return true;
}

// some level carries more than the six task dof the product kernels are built for
inline bool setup_wide_tasks(const Setup &su) {
    for (int l = 0; l < su.n_levels; l++)
        if (su.t_dof[l] > kMaxTaskDof) return true;
    return false;
}

}  // namespace dwbc
