/*
 * dwbc_oracle.h -- CPU restatement of libdwbc's per-cycle OSF/HQP torque solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under libdwbc_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, as the checker
 * and as the timed "CPU restatement of libdwbc" (kind = "port"); see oracle/README.md.
 *
 * Parity status: PINNED by the reference's binary goldens (tests/golden/cases/{1,2}, copied from
 * reference tests/cases/*) for the full-dynamics OSF cascade; see tests/test_oracle_golden.py.
 */
#ifndef DWBC_ORACLE_H
#define DWBC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXB 48   /* movable bodies */
#define ORC_MAXN 54   /* system dof */
#define ORC_MAXM 48   /* joint dof */
#define ORC_MAXCON 4  /* registered contacts */
#define ORC_MAXC 24   /* contact dof */
#define ORC_MAXL 4    /* task levels */
#define ORC_MAXTL 2   /* links per level */
#define ORC_MAXT 12   /* task dof per level */
#define ORC_MAXV 30   /* qp variables */
#define ORC_MAXR 160  /* qp rows */

typedef struct {
    int nb, ndof;
    int parent[ORC_MAXB];
    double R_T[ORC_MAXB][9];   /* child->parent rotation of the joint frame (row-major) */
    double p_T[ORC_MAXB][3];
    double axis[ORC_MAXB][3];
    double mass[ORC_MAXB];
    double com[ORC_MAXB][3];
    double inertia[ORC_MAXB][9];
} orc_model;

typedef struct {
    int n_contacts;
    int c_link[ORC_MAXCON];
    double c_point[ORC_MAXCON][3];
    double c_lx[ORC_MAXCON], c_ly[ORC_MAXCON], c_mu[ORC_MAXCON], c_muz[ORC_MAXCON];
    int n_levels;
    int t_nlinks[ORC_MAXL];
    int t_mode[ORC_MAXL][ORC_MAXTL];
    int t_link[ORC_MAXL][ORC_MAXTL];
    double t_point[ORC_MAXL][ORC_MAXTL][3];
    int has_tau_lim;
    double tau_lim[ORC_MAXM];
} orc_setup;

/* everything a test may want to look at; matrices row-major with the natural leading dim */
typedef struct {
    int status;          /* 1 = all stages ok (reference int returns ANDed) */
    int st_contact, st_task, st_redis;
    int cdof, k;
    int task_dof[ORC_MAXL];
    int qp_iter[ORC_MAXL + 1];
    int qp_nact[ORC_MAXL + 1];
    int qp_act[ORC_MAXL + 1][ORC_MAXV];
    double tau_grav[ORC_MAXM], tau_task[ORC_MAXM], tau_contact[ORC_MAXM];
    double contact_force[ORC_MAXC];
    double fstar_qp[ORC_MAXL][ORC_MAXT];
    double contact_qp[ORC_MAXL][ORC_MAXC];
    double cf_redis[ORC_MAXC];
    double G[ORC_MAXN], P_C[ORC_MAXC];
    double com[3];
} orc_out;

typedef struct {
    double A[ORC_MAXN * ORC_MAXN], A_inv[ORC_MAXN * ORC_MAXN];
    double J_C[ORC_MAXC * ORC_MAXN], Lambda_c[ORC_MAXC * ORC_MAXC], J_C_INV_T[ORC_MAXC * ORC_MAXN];
    double N_C[ORC_MAXN * ORC_MAXN], A_inv_N_C[ORC_MAXN * ORC_MAXN];
    double W[ORC_MAXM * ORC_MAXM], W_inv[ORC_MAXM * ORC_MAXM];
    double V2[ORC_MAXC * ORC_MAXM], NwJw[ORC_MAXM * ORC_MAXC];
    double CMM[6 * ORC_MAXN];
    double J_task[ORC_MAXL][ORC_MAXT * ORC_MAXN];
    double Lambda_task[ORC_MAXL][ORC_MAXT * ORC_MAXT];
    double J_kt[ORC_MAXL][ORC_MAXM * ORC_MAXT];
    double Null_task[ORC_MAXL][ORC_MAXM * ORC_MAXM];
    double qpA[ORC_MAXL + 1][ORC_MAXR * ORC_MAXV];
    double qpub[ORC_MAXL + 1][ORC_MAXR];
    int qp_rows[ORC_MAXL + 1], qp_cols[ORC_MAXL + 1];
    double link_R[ORC_MAXB][9], link_p[ORC_MAXB][3];
} orc_debug;

int orc_sizeof_model(void);
int orc_sizeof_setup(void);
int orc_sizeof_out(void);
int orc_sizeof_debug(void);

/* one control cycle: UpdateKinematics -> SetContact -> CalcContactConstraint -> CalcGravCompensation
 * -> CalcTaskControlTorque(hqp=true) -> CalcContactRedistribute(hqp=true)
 * q[ndof+1], contact flags[n_contacts], fstar = levels concatenated. dbg may be NULL. */
void orc_cycle(const orc_model *mdl, const orc_setup *su, const double *q, const uint8_t *flags,
               const double *fstar, orc_out *out, orc_debug *dbg);

/* batch driver (OpenMP over instances when built with -fopenmp).  q: B x (ndof+1), flags: B x n_contacts,
 * fstar: B x sum(task dof).  tau_out: B x 3 x m (grav, task, contact), wrench_out: B x 12 (zero padded
 * to ORC_MAXC? no: B x cdof_max where cdof_max = 6*n_contacts), status_out: B.  returns threads used. */
int orc_cycle_batch(const orc_model *mdl, const orc_setup *su, int B, const double *q, const uint8_t *flags,
                    const double *fstar, int fstar_stride, double *tau_out, double *wrench_out,
                    int32_t *status_out, int nthreads);

/* stand-alone pieces for unit tests */
int orc_solve_qp(const double *A, const double *ub, int rows, int nv, int t, int max_iter, double *x,
                 int *act, int *nact, int *iters);
int orc_pinv_cod(const double *M, int rows, int cols, double thr, double *pinv, double *V2, int *rank);

#ifdef __cplusplus
}
#endif
#endif
