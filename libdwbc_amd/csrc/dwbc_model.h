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
    int link_id(const char *name) const;       // case-insensitive, like reference src/dwbc.cpp:397-406
    void body_table(std::vector<double> &out) const;  // nb x kBodyStride device table
    void topo_table(std::vector<int> &out) const;     // parent | depth | subtree
};

// Restates RBDL's urdfreader conventions (see oracle/urdf_model.py for the list).  Returns false and fills err.
bool load_urdf(const std::string &path, bool floating_base, Model &out, std::string &err);

}  // namespace dwbc
