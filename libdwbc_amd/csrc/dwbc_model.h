// dwbc_model.h -- host-side rigid-body tree (what RobotData::LoadModelData builds through RBDL's urdfreader,
// reference src/dwbc.cpp:102-252) flattened into arrays the kernels consume.
#pragma once
#include <string>
#include <vector>

namespace dwbc {

struct Model {
    int nb = 0;    // movable bodies; body 0 is the floating base
    int ndof = 0;  // 6 + (nb - 1)
    std::vector<std::string> names;
    std::vector<int> parent, depth, subtree;
    std::vector<double> R_T;      // nb x 9  child -> parent rotation of the joint frame
    std::vector<double> p_T;      // nb x 3
    std::vector<double> axis;     // nb x 3  (unit, in the child frame)
    std::vector<double> mass;     // nb
    std::vector<double> com;      // nb x 3
    std::vector<double> inertia;  // nb x 9  about the com, body frame
    int maxdepth = 0;
    double total_mass = 0.0;

    void finalize();                           // depth / subtree / totals from `parent`
    // ---- init-time model surgery (reference RobotData::DeleteLink / AddLink / ChangeLinkToFixedJoint / ChangeLinkInertia,
    //      src/dwbc.cpp:1764-2382, 2707-2730).  Each returns false and fills err when the edit is refused.
    bool is_preorder() const;                  // bodies numbered depth first: every subtree is a contiguous index range
    bool delete_link(int link, std::string &err);  // the link AND its descendants (dwbc.cpp:1790-2047); later bodies move down
    // joint_type 0 = fixed (the body is joined to `parent`, RBDL Body::Join; no new link), 1 = revolute (a new last body).
    // R = joint frame rotation child -> parent (the reference's joint_rotm), p = its origin in the parent frame
    bool add_link(int parent_link, const char *name, int joint_type, const double *axis3, const double *R9, const double *p3, double body_mass,
                  const double *com3, const double *inertia9, std::string &err);
    bool change_link_to_fixed_joint(int link, std::string &err);  // = delete_link(link) + add_link(fixed) of the link itself (dwbc.cpp:2360-2382)
    bool change_link_inertia(int link, const double *inertia9, const double *com3, double body_mass, std::string &err);
    int link_id(const char *name) const;       // case-insensitive, like reference src/dwbc.cpp:397-406
    void body_table(std::vector<double> &out) const;  // nb x kBodyStride device table
    void topo_table(std::vector<int> &out) const;     // parent | depth | subtree
};

// Restates RBDL's urdfreader conventions (see oracle/urdf_model.py for the list).  Returns false and fills err.
bool load_urdf(const std::string &path, bool floating_base, Model &out, std::string &err);

}  // namespace dwbc
