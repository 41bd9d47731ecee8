// dwbc_model.cpp -- self-contained URDF reader + tree flattening (no libxml / urdfdom / RBDL dependency).
//
// Behaviour restated from RBDL's urdfreader (called at reference src/dwbc.cpp:115) [ext, RBDL not vendored]:
//   * children of a link are visited depth-first in ASCII order of the joint name (urdfdom keeps joints in a
//     std::map<std::string, ...>), which fixes the joint / column order of every matrix in the library;
//   * floating base = 3 translations (world axes) + spherical joint (body-frame angular velocity),
//     q = [x y z | qx qy qz | joints | qw];
//   * `fixed` joints are merged into the parent body (RBDL Body::Join): mass, com and inertia are combined;
//   * joint frame X_T = Xrot(rpy) * Xtrans(xyz), R = Rz(yaw) Ry(pitch) Rx(roll);
//   * only bodies with mass become libdwbc links (reference src/dwbc.cpp:158-203).
#include "dwbc_model.h"

#include <strings.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include "dwbc_types.h"

namespace dwbc {
namespace {

struct XmlNode {
    std::string name;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<XmlNode>> kids;
    const XmlNode *child(const char *n) const {
        for (auto &k : kids)
            if (k->name == n) return k.get();
        return nullptr;
    }
    std::string get(const char *a, const char *def = "") const {
        auto it = attr.find(a);
        return it == attr.end() ? std::string(def) : it->second;
    }
};

// minimal non-validating XML reader: elements + attributes; text, comments, PIs and DOCTYPE are skipped
class XmlReader {
  public:
    explicit XmlReader(const std::string &s) : s_(s) {}
    std::unique_ptr<XmlNode> parse(std::string &err) {
        auto root = std::make_unique<XmlNode>();
        root->name = "#doc";
        std::vector<XmlNode *> stack{root.get()};
        size_t i = 0;
        const size_t n = s_.size();
        while (i < n) {
            size_t lt = s_.find('<', i);
            if (lt == std::string::npos) break;
            if (s_.compare(lt, 4, "<!--") == 0) {
                size_t e = s_.find("-->", lt + 4);
                if (e == std::string::npos) { err = "unterminated comment"; return nullptr; }
                i = e + 3;
                continue;
            }
            if (s_.compare(lt, 2, "<?") == 0 || s_.compare(lt, 2, "<!") == 0) {
                size_t e = s_.find('>', lt);
                if (e == std::string::npos) { err = "unterminated declaration"; return nullptr; }
                i = e + 1;
                continue;
            }
            size_t gt = find_tag_end(lt);
            if (gt == std::string::npos) { err = "unterminated tag"; return nullptr; }
            std::string body = s_.substr(lt + 1, gt - lt - 1);
            i = gt + 1;
            if (!body.empty() && body[0] == '/') {
                if (stack.size() <= 1) { err = "unbalanced closing tag"; return nullptr; }
                stack.pop_back();
                continue;
            }
            bool self = !body.empty() && body.back() == '/';
            if (self) body.pop_back();
            auto node = std::make_unique<XmlNode>();
            if (!parse_tag(body, *node)) { err = "bad tag: " + body.substr(0, 40); return nullptr; }
            XmlNode *raw = node.get();
            stack.back()->kids.push_back(std::move(node));
            if (!self) stack.push_back(raw);
        }
        if (stack.size() != 1) { err = "unclosed element " + stack.back()->name; return nullptr; }
        return root;
    }

  private:
    size_t find_tag_end(size_t lt) const {
        char quote = 0;
        for (size_t i = lt + 1; i < s_.size(); i++) {
            char c = s_[i];
            if (quote) { if (c == quote) quote = 0; }
            else if (c == '"' || c == '\'') quote = c;
            else if (c == '>') return i;
        }
        return std::string::npos;
    }
    static bool parse_tag(const std::string &b, XmlNode &node) {
        size_t i = 0;
        const size_t n = b.size();
        auto ws = [&](size_t &k) { while (k < n && isspace((unsigned char)b[k])) k++; };
        ws(i);
        size_t s = i;
        while (i < n && !isspace((unsigned char)b[i])) i++;
        node.name = b.substr(s, i - s);
        if (node.name.empty()) return false;
        for (;;) {
            ws(i);
            if (i >= n) break;
            size_t ks = i;
            while (i < n && b[i] != '=' && !isspace((unsigned char)b[i])) i++;
            std::string key = b.substr(ks, i - ks);
            ws(i);
            if (i >= n || b[i] != '=') return false;
            i++;
            ws(i);
            if (i >= n || (b[i] != '"' && b[i] != '\'')) return false;
            char qc = b[i++];
            size_t vs = i;
            while (i < n && b[i] != qc) i++;
            if (i >= n) return false;
            node.attr[key] = b.substr(vs, i - vs);
            i++;
        }
        return true;
    }
    const std::string &s_;
};

struct V3 { double v[3]; };
struct M3 { double m[9]; };

bool parse_vec3(const std::string &s, double *o) {
    std::istringstream is(s);
    return bool(is >> o[0] >> o[1] >> o[2]);
}
M3 mul(const M3 &a, const M3 &b) {
    M3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
    return c;
}
M3 transpose(const M3 &a) {
    M3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[i * 3 + j] = a.m[j * 3 + i];
    return c;
}
V3 mulv(const M3 &a, const V3 &x) {
    V3 y;
    for (int i = 0; i < 3; i++) y.v[i] = a.m[i * 3] * x.v[0] + a.m[i * 3 + 1] * x.v[1] + a.m[i * 3 + 2] * x.v[2];
    return y;
}
M3 identity() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
M3 rpy_to_R(const double *rpy) {
    const double cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
    M3 Rx{{1, 0, 0, 0, cr, -sr, 0, sr, cr}}, Ry{{cp, 0, sp, 0, 1, 0, -sp, 0, cp}}, Rz{{cy, -sy, 0, sy, cy, 0, 0, 0, 1}};
    return mul(mul(Rz, Ry), Rx);
}
// I + m * skew(d) skew(d)^T
void add_parallel_axis(M3 &I, double m, const double *d) {
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) I.m[i * 3 + j] += m * ((i == j ? dd : 0.0) - d[i] * d[j]);
}

// RBDL Body::Join: body `b` of m absorbs a rigid body (mass2, com2, I2 about its com) attached through the frame (R child -> parent, p)
void join_body(Model &m, int b, const M3 &R, const V3 &p, double mass2, const V3 &com2, const M3 &I2_in) {
    if (mass2 == 0.0) return;
    V3 c2 = mulv(R, com2);
    for (int a = 0; a < 3; a++) c2.v[a] += p.v[a];
    M3 I2 = mul(mul(R, I2_in), transpose(R));
    const double m1 = m.mass[b], mt = m1 + mass2;
    double c1[3], c[3], d1[3], d2[3];
    for (int a = 0; a < 3; a++) {
        c1[a] = m.com[b * 3 + a];
        c[a] = (m1 * c1[a] + mass2 * c2.v[a]) / mt;
        d1[a] = c1[a] - c[a];
        d2[a] = c2.v[a] - c[a];
    }
    M3 I1;
    for (int a = 0; a < 9; a++) I1.m[a] = m.inertia[b * 9 + a];
    add_parallel_axis(I1, m1, d1);
    add_parallel_axis(I2, mass2, d2);
    for (int a = 0; a < 9; a++) m.inertia[b * 9 + a] = I1.m[a] + I2.m[a];
    for (int a = 0; a < 3; a++) m.com[b * 3 + a] = c[a];
    m.mass[b] = mt;
}
template <class T>
void erase_rows(std::vector<T> &v, int first, int count, int width) {
    v.erase(v.begin() + (size_t)first * width, v.begin() + (size_t)(first + count) * width);
}

struct UrdfLink { double mass = 0; V3 com{{0, 0, 0}}; M3 inertia{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; };
struct UrdfJoint { std::string name, type, parent, child; V3 xyz{{0, 0, 0}}; M3 R = identity(); V3 axis{{1, 0, 0}}; };

struct Builder {
    std::map<std::string, UrdfLink> links;
    std::map<std::string, UrdfJoint> joints;  // std::map: ASCII order of joint names
    std::map<std::string, std::vector<std::string>> children;
    Model *out;
    std::string err;

    int add_body(const std::string &name, int parent, const M3 &R, const V3 &p, const V3 &axis, const UrdfLink &l) {
        Model &m = *out;
        m.names.push_back(name);
        m.parent.push_back(parent);
        m.R_T.insert(m.R_T.end(), R.m, R.m + 9);
        m.p_T.insert(m.p_T.end(), p.v, p.v + 3);
        m.axis.insert(m.axis.end(), axis.v, axis.v + 3);
        m.mass.push_back(l.mass);
        m.com.insert(m.com.end(), l.com.v, l.com.v + 3);
        m.inertia.insert(m.inertia.end(), l.inertia.m, l.inertia.m + 9);
        return (int)m.names.size() - 1;
    }
    bool visit(const std::string &link, int body, const M3 &Racc, const V3 &pacc) {
        for (const std::string &jn : children[link]) {
            const UrdfJoint &j = joints[jn];
            M3 Rj = mul(Racc, j.R);
            V3 t = mulv(Racc, j.xyz), pj;
            for (int a = 0; a < 3; a++) pj.v[a] = pacc.v[a] + t.v[a];
            auto lit = links.find(j.child);
            if (lit == links.end()) { err = "joint " + jn + " refers to unknown link " + j.child; return false; }
            const UrdfLink &l = lit->second;
            if (j.type == "fixed") {
                Model &m = *out;
                join_body(m, body, Rj, pj, l.mass, l.com, l.inertia);  // Body::Join
                if (!visit(j.child, body, Rj, pj)) return false;
            } else if (j.type == "revolute" || j.type == "continuous") {
                double nrm = sqrt(j.axis.v[0] * j.axis.v[0] + j.axis.v[1] * j.axis.v[1] + j.axis.v[2] * j.axis.v[2]);
                if (nrm == 0.0) { err = "joint " + jn + " has a zero axis"; return false; }
                V3 ax{{j.axis.v[0] / nrm, j.axis.v[1] / nrm, j.axis.v[2] / nrm}};
                int nbdy = add_body(j.child, body, Rj, pj, ax, l);
                if (!visit(j.child, nbdy, identity(), V3{{0, 0, 0}})) return false;
            } else {
                err = "joint " + jn + ": unsupported type '" + j.type + "' (revolute, continuous, fixed)";
                return false;
            }
        }
        return true;
    }
};

}  // namespace

void Model::finalize() {
    nb = (int)parent.size();
    ndof = 6 + nb - 1;
    depth.assign(nb, 0);
    subtree.assign(nb, 1);
    maxdepth = 0;
    total_mass = 0.0;
    for (int i = 0; i < nb; i++) {
        if (i > 0) depth[i] = depth[parent[i]] + 1;
        maxdepth = std::max(maxdepth, depth[i]);
        total_mass += mass[i];
    }
    for (int i = nb - 1; i > 0; i--) subtree[parent[i]] += subtree[i];
}

bool Model::is_preorder() const {
    // pre-order <=> the parent of body i is body i-1 or one of its ancestors
    for (int i = 1; i < (int)parent.size(); i++) {
        if (parent[i] < 0 || parent[i] >= i) return false;
        int a = i - 1;
        while (a != parent[i] && a > 0) a = parent[a];
        if (a != parent[i]) return false;
    }
    return true;
}


bool Model::delete_link(int link, std::string &err) {
    if (link <= 0 || link >= nb) { err = "DeleteLink: bad link (the base link cannot be deleted)"; return false; }
    if (!is_preorder()) { err = "DeleteLink: model is not numbered depth first"; return false; }
    const int len = subtree[link];
    erase_rows(names, link, len, 1);
    erase_rows(parent, link, len, 1);
    erase_rows(R_T, link, len, 9);
    erase_rows(p_T, link, len, 3);
    erase_rows(axis, link, len, 3);
    erase_rows(mass, link, len, 1);
    erase_rows(com, link, len, 3);
    erase_rows(inertia, link, len, 9);
    for (size_t i = 0; i < parent.size(); i++)
        if (parent[i] >= link + len) parent[i] -= len;  // (no surviving body has its parent inside the deleted range)
    finalize();
    return true;
}

bool Model::add_link(int parent_link, const char *name, int joint_type, const double *axis3, const double *R9, const double *p3, double body_mass,
                     const double *com3, const double *inertia9, std::string &err) {
    if (parent_link < 0 || parent_link >= nb) { err = "AddLink: bad parent link"; return false; }
    M3 R, I;
    V3 p, c;
    for (int a = 0; a < 9; a++) { R.m[a] = R9[a]; I.m[a] = inertia9[a]; }
    for (int a = 0; a < 3; a++) { p.v[a] = p3[a]; c.v[a] = com3[a]; }
    if (joint_type == 0) {
        join_body(*this, parent_link, R, p, body_mass, c, I);
        finalize();
        return true;
    }
    if (joint_type != 1) { err = "AddLink: joint type must be fixed (0) or revolute (1)"; return false; }
    // the new body becomes the LAST one (RBDL appends; its joint is the last generalized coordinate).  The kernels need the
    // depth-first numbering to survive that: the parent must be the last body or one of its ancestors
    int a = nb - 1;
    while (a != parent_link && a > 0) a = parent[a];
    if (a != parent_link) {
        err = "AddLink: a revolute link can only be attached to the last link or one of its ancestors (the bodies stay numbered depth first; "
              "the new joint is the last coordinate, as in the reference)";
        return false;
    }
    const double nrm = sqrt(axis3[0] * axis3[0] + axis3[1] * axis3[1] + axis3[2] * axis3[2]);
    if (nrm == 0.0) { err = "AddLink: zero joint axis"; return false; }
    names.push_back(name ? name : "");
    parent.push_back(parent_link);
    R_T.insert(R_T.end(), R.m, R.m + 9);
    p_T.insert(p_T.end(), p.v, p.v + 3);
    for (int k = 0; k < 3; k++) axis.push_back(axis3[k] / nrm);
    mass.push_back(body_mass);
    com.insert(com.end(), c.v, c.v + 3);
    inertia.insert(inertia.end(), I.m, I.m + 9);
    finalize();
    return true;
}

bool Model::change_link_to_fixed_joint(int link, std::string &err) {
    if (link <= 0 || link >= nb) { err = "ChangeLinkToFixedJoint: bad link"; return false; }
    const int par = parent[link];
    M3 R, I;
    V3 p, c;
    for (int a = 0; a < 9; a++) { R.m[a] = R_T[link * 9 + a]; I.m[a] = inertia[link * 9 + a]; }
    for (int a = 0; a < 3; a++) { p.v[a] = p_T[link * 3 + a]; c.v[a] = com[link * 3 + a]; }
    const double ms = mass[link];
    const std::string nm = names[link];
    if (!delete_link(link, err)) return false;  // descendants go with it, as in the reference
    const double ax[3] = {0, 0, 1};
    return add_link(par, nm.c_str(), 0, ax, R.m, p.v, ms, c.v, I.m, err);
}

bool Model::change_link_inertia(int link, const double *inertia9, const double *com3, double body_mass, std::string &err) {
    if (link < 0 || link >= nb) { err = "ChangeLinkInertia: bad link"; return false; }
    for (int a = 0; a < 9; a++) inertia[link * 9 + a] = inertia9[a];
    for (int a = 0; a < 3; a++) com[link * 3 + a] = com3[a];
    mass[link] = body_mass;
    finalize();
    return true;
}

int Model::link_id(const char *name) const {
    for (int i = 0; i < nb; i++)
        if (strcasecmp(names[i].c_str(), name) == 0) return i;
    if (strcasecmp(name, "COM") == 0) return nb;  // the synthetic centre-of-mass link, id = link_num_ (reference src/dwbc.cpp:230-231)
    return -1;
}

void Model::body_table(std::vector<double> &out) const {
    out.assign((size_t)nb * kBodyStride, 0.0);
    for (int i = 0; i < nb; i++) {
        double *o = out.data() + (size_t)i * kBodyStride;
        for (int a = 0; a < 9; a++) o[BF_RT + a] = R_T[i * 9 + a];
        for (int a = 0; a < 3; a++) {
            o[BF_PT + a] = p_T[i * 3 + a];
            o[BF_AXIS + a] = axis[i * 3 + a];
            o[BF_COM + a] = com[i * 3 + a];
        }
        o[BF_MASS] = mass[i];
        const double *I = inertia.data() + i * 9;
        o[BF_ICOM + 0] = I[0]; o[BF_ICOM + 1] = I[1]; o[BF_ICOM + 2] = I[2];
        o[BF_ICOM + 3] = I[4]; o[BF_ICOM + 4] = I[5]; o[BF_ICOM + 5] = I[8];
    }
}

void Model::topo_table(std::vector<int> &out) const {
    out.resize((size_t)3 * nb);
    for (int i = 0; i < nb; i++) {
        out[i] = parent[i] < 0 ? 0 : parent[i];
        out[nb + i] = depth[i];
        out[2 * nb + i] = subtree[i];
    }
    // the coupled dof pairs of the mass matrix, (j << 8) | k for every dof k on the path from dof j to the root (k <= j;
    // dofs 0..5 = floating base, dof d >= 6 = joint of body d - 5): the CRBA fills A[j][k] = A[k][j] = S_k . F_j pair by
    // pair, 64 independent pairs per round, instead of one dependent walk up the tree per dof
    std::vector<int> pairs;
    const int nd = nb + 5;
    for (int j = 0; j < nd; j++) {
        int k = j;
        for (;;) {
            pairs.push_back((j << 8) | k);
            if (k == 0) break;
            if (k < 6) {
                k--;
            } else {
                const int pb = parent[k - 5] < 0 ? 0 : parent[k - 5];
                k = pb == 0 ? 5 : pb + 5;
            }
        }
    }
    out.push_back((int)pairs.size());
    out.insert(out.end(), pairs.begin(), pairs.end());
}

bool load_urdf(const std::string &path, bool floating_base, Model &out, std::string &err) {
    if (!floating_base) { err = "only floating-base models are supported (reference call sites use floating=true)"; return false; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    XmlReader rd(text);
    auto doc = rd.parse(err);
    if (!doc) return false;
    const XmlNode *robot = doc->child("robot");
    if (!robot) { err = "no <robot> element"; return false; }
    out = Model();
    Builder b;
    b.out = &out;
    std::map<std::string, bool> is_child;
    for (auto &k : robot->kids) {
        if (k->name == "link") {
            UrdfLink l;
            if (const XmlNode *in = k->child("inertial")) {
                double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0};
                if (const XmlNode *o = in->child("origin")) {
                    if (!o->get("xyz").empty()) parse_vec3(o->get("xyz"), xyz);
                    if (!o->get("rpy").empty()) parse_vec3(o->get("rpy"), rpy);
                }
                if (const XmlNode *mm = in->child("mass")) l.mass = atof(mm->get("value", "0").c_str());
                M3 I{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
                if (const XmlNode *it = in->child("inertia")) {
                    auto g = [&](const char *a) { return atof(it->get(a, "0").c_str()); };
                    I = M3{{g("ixx"), g("ixy"), g("ixz"), g("ixy"), g("iyy"), g("iyz"), g("ixz"), g("iyz"), g("izz")}};
                }
                M3 Ri = rpy_to_R(rpy);
                l.inertia = mul(mul(Ri, I), transpose(Ri));
                for (int a = 0; a < 3; a++) l.com.v[a] = xyz[a];
            }
            b.links[k->get("name")] = l;
            b.children[k->get("name")];
        }
    }
    for (auto &k : robot->kids) {
        if (k->name != "joint") continue;
        UrdfJoint j;
        j.name = k->get("name");
        j.type = k->get("type");
        const XmlNode *p = k->child("parent"), *c = k->child("child");
        if (!p || !c) { err = "joint " + j.name + " lacks parent/child"; return false; }
        j.parent = p->get("link");
        j.child = c->get("link");
        double rpy[3] = {0, 0, 0};
        if (const XmlNode *o = k->child("origin")) {
            if (!o->get("xyz").empty()) parse_vec3(o->get("xyz"), j.xyz.v);
            if (!o->get("rpy").empty()) parse_vec3(o->get("rpy"), rpy);
        }
        j.R = rpy_to_R(rpy);
        if (const XmlNode *a = k->child("axis")) parse_vec3(a->get("xyz", "1 0 0"), j.axis.v);
        if (!b.links.count(j.parent)) { err = "joint " + j.name + " refers to unknown link " + j.parent; return false; }
        b.joints[j.name] = j;
        b.children[j.parent].push_back(j.name);
        is_child[j.child] = true;
    }
    for (auto &kv : b.children) std::sort(kv.second.begin(), kv.second.end());
    std::string root;
    int nroots = 0;
    for (auto &kv : b.links)
        if (!is_child.count(kv.first)) { root = kv.first; nroots++; }
    if (nroots != 1) { err = "URDF must have exactly one root link"; return false; }
    b.add_body(root, -1, identity(), V3{{0, 0, 0}}, V3{{0, 0, 0}}, b.links[root]);
    if (!b.visit(root, 0, identity(), V3{{0, 0, 0}})) { err = b.err; return false; }
    out.finalize();
    if (out.nb > kMaxBodies) { err = "too many bodies"; return false; }
    return true;
}

}  // namespace dwbc
