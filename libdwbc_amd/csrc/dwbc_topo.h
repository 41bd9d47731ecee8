// Compile-time kinematic trees.  The joint-space mass matrix is sparse by the tree (A[i][j] != 0 only when dof i is an
// ancestor or a descendant of dof j); a kernel instantiated for one robot size can be given that pattern as a constant and
// drop the structurally zero work from its A^-1 sweep (sweep_inverse_tree in dwbc_cycle2.h).  The host compares the loaded
// model's parent table with the constant one (setup_set_parents) and the kernel falls back to the dense sweep on a mismatch,
// so a table here is an optimisation for a known robot, never an assumption about the model.
#pragma once

namespace dwbc {

// no constant tree: the dense sweep, correct for any model of the instantiated size
struct TopoGeneric {};

// every pair couples: sweep_inverse_tree<TopoDense<NN>, NN> is the dense sweep with compile-time pivots
template <int NN>
struct TopoDense {
    static constexpr int ndof = NN;
    static constexpr unsigned long long relatives(int) { return ~0ull; }
};

// TOCABI (libdwbc_amd/data/dyros_tocabi.urdf; the robot of every reference test, example and BASELINE config): pelvis 0, left leg
// 1-6, right leg 7-12, waist 13-15, left arm 16-23, head 24-25, right arm 26-33.  RBDL body order = URDF depth-first order.
struct TopoTocabi {
    static constexpr int nb = 34;
    static constexpr int ndof = nb + 5;
    static constexpr int maxdepth = 11;  // deepest bodies: the hands (3 waist joints + 8 arm joints below the pelvis)
    static constexpr int parent[nb] = {0, 0, 1, 2, 3, 4, 5, 0, 7, 8, 9, 10, 11, 0, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 15, 24, 15, 26, 27, 28, 29, 30, 31, 32};
    static constexpr bool anc_or_self(int a, int b) {
        while (b > a) b = parent[b];
        return a == b;
    }
    // bit i set <=> dof i couples with dof k (dofs 0..5 belong to body 0, dof d >= 6 to body d - 5)
    static constexpr unsigned long long relatives(int k) {
        unsigned long long m = 0;
        const int bk = k < 6 ? 0 : k - 5;
        for (int i = 0; i < ndof; i++) {
            const int bi = i < 6 ? 0 : i - 5;
            if (anc_or_self(bi, bk) || anc_or_self(bk, bi)) m |= 1ull << i;
        }
        return m;
    }
    static constexpr int depth_of(int b) {
        int d = 0;
        while (b > 0) { b = parent[b]; d++; }
        return d;
    }
    static constexpr int computed_maxdepth() {
        int m = 0;
        for (int b = 0; b < nb; b++) m = depth_of(b) > m ? depth_of(b) : m;
        return m;
    }
    static bool matches(int n_bodies, const int *par) {
        if (n_bodies != nb) return false;
        for (int i = 0; i < nb; i++)
            if ((par[i] < 0 ? 0 : par[i]) != parent[i]) return false;
        return true;
    }
};

static_assert(TopoTocabi::computed_maxdepth() == TopoTocabi::maxdepth, "TopoTocabi::maxdepth");

// A kernel pack built for ONE model's tree (dwbc_pack.hip with -DDWBC_PACK_PARENTS=p0,p1,...: the parent of every body, body 0's
// entry 0): the same compile-time sparsity and round counts TopoTocabi gives the built-in kernels.  The loader uses such a pack
// only for a model whose parent table equals this one (dwbc_pack_parents).
#ifdef DWBC_PACK_PARENTS
namespace pack_tree {
constexpr int kParent[DWBC_PACK_NB] = {DWBC_PACK_PARENTS};
constexpr int max_depth() {
    int m = 0;
    for (int b = 0; b < DWBC_PACK_NB; b++) {
        int d = 0;
        for (int c = b; c > 0; c = kParent[c]) d++;
        m = d > m ? d : m;
    }
    return m;
}
}  // namespace pack_tree
struct TopoPack {
    static constexpr int nb = DWBC_PACK_NB;
    static constexpr int ndof = nb + 5;
    static constexpr int parent[nb] = {DWBC_PACK_PARENTS};
    static constexpr bool anc_or_self(int a, int b) {
        while (b > a) b = parent[b];
        return a == b;
    }
    static constexpr unsigned long long relatives(int k) {
        unsigned long long m = 0;
        const int bk = k < 6 ? 0 : k - 5;
        for (int i = 0; i < ndof; i++) {
            const int bi = i < 6 ? 0 : i - 5;
            if (anc_or_self(bi, bk) || anc_or_self(bk, bi)) m |= 1ull << i;
        }
        return m;
    }
    static constexpr int maxdepth = pack_tree::max_depth();
};
#endif

}  // namespace dwbc
