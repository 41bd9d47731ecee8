// dwbc_model_capi.cpp -- C-ABI of the init-time model surgery (include/dwbc_batch.h, section "model surgery"): host code only.
// Every entry point returns a NEW model handle and leaves its argument untouched: batches keep a pointer to the model they
// were created with, and a model of another size needs other kernels (a pack, dwbc_pack.hip) anyway.
#include <cstring>
#include <string>

#include "dwbc_capi_internal.h"

using namespace dwbc;

namespace {
dwbc_model *finish(dwbc_model *mm, bool ok, const std::string &err) {
    if (!ok) {
        capi_err() = err;
        delete mm;
        return nullptr;
    }
    return mm;
}
}  // namespace

extern "C" {

dwbc_model *dwbc_model_delete_link(const dwbc_model *m, int link) {
    if (!m) { capi_err() = "NULL model"; return nullptr; }
    auto *mm = new dwbc_model(*m);
    std::string err;
    return finish(mm, mm->m.delete_link(link, err), err);
}

dwbc_model *dwbc_model_add_link(const dwbc_model *m, int parent_link, const char *link_name, int joint_type, const double *joint_axis, const double *joint_rotm,
                                const double *joint_trans, double body_mass, const double *com_position, const double *inertia) {
    if (!m || !joint_axis || !joint_rotm || !joint_trans || !com_position || !inertia) { capi_err() = "NULL argument"; return nullptr; }
    auto *mm = new dwbc_model(*m);
    std::string err;
    return finish(mm, mm->m.add_link(parent_link, link_name, joint_type, joint_axis, joint_rotm, joint_trans, body_mass, com_position, inertia, err), err);
}

dwbc_model *dwbc_model_change_link_to_fixed_joint(const dwbc_model *m, int link) {
    if (!m) { capi_err() = "NULL model"; return nullptr; }
    auto *mm = new dwbc_model(*m);
    std::string err;
    return finish(mm, mm->m.change_link_to_fixed_joint(link, err), err);
}

dwbc_model *dwbc_model_change_link_inertia(const dwbc_model *m, int link, const double *com_inertia, const double *com_position, double com_mass) {
    if (!m || !com_inertia || !com_position) { capi_err() = "NULL argument"; return nullptr; }
    auto *mm = new dwbc_model(*m);
    std::string err;
    return finish(mm, mm->m.change_link_inertia(link, com_inertia, com_position, com_mass, err), err);
}

}  // extern "C"
