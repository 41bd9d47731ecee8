"""URDF -> flat rigid-body-tree arrays, restating the conventions of RBDL's urdfreader.

TEST INFRASTRUCTURE ONLY (part of oracle/).  The product has its own C++ URDF reader in
libdwbc_amd/csrc/dwbc_model.cpp; tests compare the two.

The reference loads its model with RigidBodyDynamics::Addons::URDFReadFromFile
(reference src/dwbc.cpp:115).  RBDL (saga0619/rbdl-orb, unpinned) is NOT in /root/reference, so
this file restates the published behaviour of that reader [ext]:

  * children of a link are visited depth-first in ASCII order of the *joint name*
    (urdfdom keeps joints in a std::map keyed by name),
  * a floating base is a 3-DoF translation (world axes) followed by a spherical joint whose
    velocity is the body-frame angular velocity; q = [x y z | qx qy qz | joints... | qw],
  * `fixed` joints are merged: the child's inertia is joined into the parent body
    (RBDL Body::Join), no DoF is created,
  * joint frame  X_T = Xrot(rpy) * Xtrans(xyz)   (R = Rz(y) Ry(p) Rx(r), child -> parent),
  * only bodies with non-zero mass become libdwbc "links" (reference src/dwbc.cpp:158-203);
    for TOCABI that is every movable body, so link id == body id here.

Pinned by: tests/golden/cases/{1,2}/A_inv_ and J_C (joint order, frames, inertias).
"""
import json
import xml.etree.ElementTree as ET

import numpy as np


def _vec(s, n=3):
    v = [float(x) for x in s.split()]
    assert len(v) == n
    return np.array(v)


def rpy_to_R(rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _join(m1, c1, I1, m2, c2, I2):
    """Combine two rigid bodies given in the same frame (mass, com, inertia about own com)."""
    m = m1 + m2
    c = (m1 * c1 + m2 * c2) / m
    d1, d2 = c1 - c, c2 - c
    I = I1 + m1 * (_skew(d1) @ _skew(d1).T) + I2 + m2 * (_skew(d2) @ _skew(d2).T)
    return m, c, I


def load_urdf(path):
    root = ET.parse(path).getroot()
    links = {}
    for l in root.findall("link"):
        name = l.get("name")
        ine = l.find("inertial")
        if ine is None:
            links[name] = dict(mass=0.0, com=np.zeros(3), inertia=np.zeros((3, 3)))
            continue
        org = ine.find("origin")
        xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
        rpy = _vec(org.get("rpy", "0 0 0")) if org is not None else np.zeros(3)
        m = float(ine.find("mass").get("value"))
        it = ine.find("inertia")
        I = np.array(
            [
                [float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))],
                [float(it.get("ixy")), float(it.get("iyy")), float(it.get("iyz"))],
                [float(it.get("ixz")), float(it.get("iyz")), float(it.get("izz"))],
            ]
        )
        Ri = rpy_to_R(rpy)
        links[name] = dict(mass=m, com=xyz, inertia=Ri @ I @ Ri.T)
    joints = {}
    children = {n: [] for n in links}
    child_names = set()
    for j in root.findall("joint"):
        name = j.get("name")
        org = j.find("origin")
        xyz = _vec(org.get("xyz", "0 0 0")) if org is not None else np.zeros(3)
        rpy = _vec(org.get("rpy", "0 0 0")) if org is not None else np.zeros(3)
        ax = j.find("axis")
        axis = _vec(ax.get("xyz")) if ax is not None else np.array([1.0, 0, 0])
        jd = dict(
            name=name,
            type=j.get("type"),
            parent=j.find("parent").get("link"),
            child=j.find("child").get("link"),
            xyz=xyz,
            R=rpy_to_R(rpy),
            axis=axis,
        )
        joints[name] = jd
        children[jd["parent"]].append(name)
        child_names.add(jd["child"])
    roots = [n for n in links if n not in child_names]
    assert len(roots) == 1, roots
    for n in children:
        children[n].sort()  # std::map<string,...> order == ASCII order of joint names

    bodies = []  # dicts: name parent R_T p_T axis mass com inertia

    def add_body(name, parent, R_T, p_T, axis, ld):
        bodies.append(
            dict(name=name, parent=parent, R_T=R_T, p_T=p_T, axis=axis, mass=ld["mass"], com=ld["com"].copy(), inertia=ld["inertia"].copy())
        )
        return len(bodies) - 1

    def visit(link_name, body_idx, R_acc, p_acc):
        """R_acc,p_acc: pose of `link_name` frame in movable body `body_idx` frame."""
        for jn in children[link_name]:
            jd = joints[jn]
            R_j = R_acc @ jd["R"]
            p_j = p_acc + R_acc @ jd["xyz"]
            ld = links[jd["child"]]
            if jd["type"] == "fixed":
                b = bodies[body_idx]
                if ld["mass"] != 0.0:
                    c2 = p_j + R_j @ ld["com"]
                    I2 = R_j @ ld["inertia"] @ R_j.T
                    b["mass"], b["com"], b["inertia"] = _join(b["mass"], b["com"], b["inertia"], ld["mass"], c2, I2)
                visit(jd["child"], body_idx, R_j, p_j)
            elif jd["type"] in ("revolute", "continuous"):
                nb = add_body(jd["child"], body_idx, R_j, p_j, jd["axis"] / np.linalg.norm(jd["axis"]), ld)
                visit(jd["child"], nb, np.eye(3), np.zeros(3))
            else:
                raise NotImplementedError(jd["type"])

    add_body(roots[0], -1, np.eye(3), np.zeros(3), np.zeros(3), links[roots[0]])
    visit(roots[0], 0, np.eye(3), np.zeros(3))
    nb = len(bodies)
    model = dict(
        nb=nb,
        ndof=6 + nb - 1,
        names=[b["name"] for b in bodies],
        parent=np.array([b["parent"] for b in bodies], dtype=np.int32),
        R_T=np.array([b["R_T"] for b in bodies]),
        p_T=np.array([b["p_T"] for b in bodies]),
        axis=np.array([b["axis"] for b in bodies]),
        mass=np.array([b["mass"] for b in bodies]),
        com=np.array([b["com"] for b in bodies]),
        inertia=np.array([b["inertia"] for b in bodies]),
    )
    return model


def model_to_json(model, path):
    out = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in model.items()}
    with open(path, "w") as f:
        json.dump(out, f)


def model_from_json(path):
    with open(path) as f:
        d = json.load(f)
    m = dict(d)
    for k in ("R_T", "p_T", "axis", "mass", "com", "inertia"):
        m[k] = np.array(d[k], dtype=np.float64)
    m["parent"] = np.array(d["parent"], dtype=np.int32)
    return m


if __name__ == "__main__":
    import sys

    m = load_urdf(sys.argv[1])
    print(m["nb"], m["ndof"], m["mass"].sum())
    for i, n in enumerate(m["names"]):
        print(i, n, m["parent"][i], m["mass"][i])
    if len(sys.argv) > 2:
        model_to_json(m, sys.argv[2])
