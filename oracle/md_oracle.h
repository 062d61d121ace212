/* md_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, double-precision CPU restatement of HOOMD-blue 4.1.0's short-range MD step
 * (cell list -> binned neighbor list -> pair force -> velocity-Verlet half steps), written
 * from the reference's CPU algorithm.  Every function cites the reference file:line it
 * follows (paths relative to the HOOMD-blue source tree).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this code,
 * and only as the checker / reported CPU baseline.  The shipped GPU path never links,
 * imports or executes anything in this directory.
 *
 * Parity pinning: the evaluator, box arithmetic and indexers are checked against golden
 * vectors produced by oracle/ref_harness.cc, which compiles the reference's own headers
 * (hoomd/BoxDim.h, hoomd/Index1D.h, hoomd/md/EvaluatorPairLJ.h,
 * example_plugins/pair_plugin/EvaluatorPairExample.h) where they lie, and against the
 * reference's own fixtures (forces_and_energies.json LJ rows, true_pair_list.json, the
 * hand-built cases of hoomd/test/test_cell_list.cc and hoomd/md/test/test_neighborlist.cc).
 */
#ifndef MD_ORACLE_H
#define MD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hoomd/BoxDim.h:633-641 (member layout restated as plain doubles) */
typedef struct
    {
    double lo[3];
    double hi[3];
    double L[3];
    double Linv[3];
    double xy, xz, yz;
    unsigned char periodic[3];
    } orc_box_t;

typedef struct
    {
    double x, y, z, w;
    } orc_d4;

typedef struct
    {
    double x, y, z;
    } orc_d3;

typedef struct
    {
    int x, y, z;
    } orc_i3;

typedef struct
    {
    unsigned int x, y;
    } orc_u2;

#define ORC_NO_BODY 0xffffffffu

/* evaluator ids */
#define ORC_EV_LJ 0      /* hoomd/md/EvaluatorPairLJ.h */
#define ORC_EV_EXAMPLE 1 /* example_plugins/pair_plugin/EvaluatorPairExample.h */

/* shift modes, hoomd/md/PotentialPair.h:126-131 */
#define ORC_SHIFT_NONE 0
#define ORC_SHIFT_SHIFT 1
#define ORC_SHIFT_XPLOR 2

/* ---- box ---- */
void orc_box_init(orc_box_t* box, double Lx, double Ly, double Lz, double xy, double xz, double yz);
void orc_box_set_lo_hi(orc_box_t* box, const double lo[3], const double hi[3]);
void orc_box_nearest_plane_distance(const orc_box_t* box, double out[3]);
void orc_box_make_fraction(const orc_box_t* box, const double v[3], const double ghost_width[3], double out[3]);
void orc_box_min_image(const orc_box_t* box, double v[3]);      /* CPU (branch) form */
void orc_box_min_image_rint(const orc_box_t* box, double v[3]); /* GPU (rint) form   */
void orc_box_wrap(const orc_box_t* box, double v[3], int img[3]);

/* ---- indexers (hoomd/Index1D.h) ---- */
unsigned int orc_index2d(unsigned int w, unsigned int i, unsigned int j);
unsigned int orc_index3d(unsigned int w, unsigned int h, unsigned int i, unsigned int j, unsigned int k);

/* ---- cell list ---- */
void orc_cell_dimensions(const orc_box_t* box,
                         double nominal_width,
                         unsigned int multiple,
                         const double ghost_width[3],
                         unsigned int ndim,
                         unsigned int dim_out[3]);
unsigned int orc_cell_n_adj(const unsigned int dim[3], unsigned int radius, unsigned int ndim);
void orc_cell_adj(const unsigned int dim[3], unsigned int radius, unsigned int ndim, unsigned int* adj_out);
/* flag_mode: 0 = particle index in xyzf.w, 1 = charge, 2 = type */
void orc_compute_cell_list(const orc_d4* pos,
                           const double* charge,
                           const unsigned int* body,
                           unsigned int N,
                           unsigned int n_ghost,
                           const orc_box_t* box,
                           const unsigned int dim[3],
                           unsigned int Nmax,
                           const double ghost_width[3],
                           int flag_mode,
                           unsigned int* cell_size,
                           orc_d4* xyzf,
                           orc_u2* type_body,
                           unsigned int* cell_idx,
                           unsigned int conditions[3]);

/* ---- neighbor list ---- */
void orc_update_rlist(const double* r_cut,
                      unsigned int ntypes,
                      double r_buff,
                      double* r_listsq,
                      double* rcut_max,
                      double* rcut_max_max,
                      double* rcut_min);
size_t orc_build_head_list(const orc_d4* pos, unsigned int N, const unsigned int* Nmax, size_t* head_list);
int orc_nlist_check_conditions(const unsigned int* conditions, unsigned int* Nmax, unsigned int ntypes);
/* storage_full: 1 = full list, 0 = half list (i<j) */
void orc_nlist_binned(const orc_d4* pos,
                      const unsigned int* body,
                      unsigned int N,
                      const orc_box_t* box,
                      const unsigned int dim[3],
                      const double ghost_width[3],
                      const unsigned int* cell_size,
                      const orc_d4* cell_xyzf,
                      const unsigned int* cell_adj,
                      unsigned int n_adj,
                      unsigned int cell_Nmax,
                      const double* r_cut,
                      const double* r_listsq,
                      unsigned int ntypes,
                      int storage_full,
                      int filter_body,
                      const size_t* head_list,
                      const unsigned int* Nmax,
                      unsigned int* nlist,
                      unsigned int* n_neigh,
                      unsigned int* conditions);
int orc_nlist_distance_check(const orc_d4* pos,
                             const orc_d4* last_pos,
                             unsigned int N,
                             const orc_box_t* box,
                             const double* rcut_max,
                             double r_buff,
                             const double lambda[3]);
void orc_nlist_set_last_pos(const orc_d4* pos, unsigned int N, orc_d4* last_pos);
void orc_nlist_filter(unsigned int N,
                      const size_t* head_list,
                      const unsigned int* n_ex_idx,
                      const unsigned int* ex_list_idx,
                      unsigned int ex_pitch,
                      unsigned int* n_neigh,
                      unsigned int* nlist);

/* hoomd/md/NeighborList.cc:1325-1418 state machine */
typedef struct
    {
    uint64_t last_updated_tstep;
    uint64_t last_checked_tstep;
    int last_check_result;
    int force_update;
    uint64_t rebuild_check_delay;
    int dist_check;
    double r_buff;
    uint64_t updates, forced_updates, dangerous_updates;
    uint64_t update_periods[100];
    } orc_nlist_state_t;
void orc_nlist_state_init(orc_nlist_state_t* s, double r_buff);
/* returns 1 when a distance check is required to finish the decision (caller then calls
   orc_needs_updating_finish with the check's result), 0 when *result is final. */
int orc_needs_updating_begin(orc_nlist_state_t* s, uint64_t timestep, int* result, int* dangerous);
int orc_needs_updating_finish(orc_nlist_state_t* s, uint64_t timestep, int check_result, int dangerous);
unsigned int orc_nlist_smallest_rebuild(const orc_nlist_state_t* s);

/* ---- pair potential ---- */
/* params: 2 doubles per type pair. LJ: {sigma_6, epsilon_x_4}; Example: {k, sigma}. */
int orc_eval_pair(int evaluator,
                  double rsq,
                  double rcutsq,
                  const double* params2,
                  int energy_shift,
                  double* force_divr,
                  double* pair_eng);
void orc_eval_lrc(int evaluator, double rcutsq, const double* params2, double* energy_integral, double* pressure_integral);
void orc_pair_forces(int evaluator,
                     const orc_d4* pos,
                     unsigned int N,
                     unsigned int n_total, /* N + ghosts: size of force/virial arrays */
                     const orc_box_t* box,
                     const unsigned int* n_neigh,
                     const unsigned int* nlist,
                     const size_t* head_list,
                     const double* params, /* 2*ntypes^2 */
                     const double* rcutsq,
                     const double* ronsq,
                     unsigned int ntypes,
                     int shift_mode,
                     int third_law,
                     int compute_virial,
                     orc_d4* force,
                     double* virial,
                     size_t virial_pitch);

/* ---- bonds (hoomd/md/EvaluatorBondFENE.h, hoomd/md/PotentialBond.h) ---- */
/* params5: {k, r_0, sigma_6, epsilon_x_4, delta} (fene_params, EvaluatorBondFENE.h:30-69) */
int orc_eval_bond_fene(double rsq, const double* params5, double* force_divr, double* bond_eng);
/* Returns 0, or 1 + the number of the first bond whose energy is undefined (the reference throws
   "bond out of bounds" there, PotentialBond.h:330-336).  bonds: 2 tags per bond; bond_type per bond. */
int orc_bond_forces_fene(const orc_d4* pos,
                         const unsigned int* rtag,
                         unsigned int N,
                         unsigned int n_total,
                         const orc_box_t* box,
                         const unsigned int* bonds,
                         const unsigned int* bond_type,
                         unsigned int n_bonds,
                         const double* params, /* 5 per bond type */
                         int compute_virial,
                         orc_d4* force,
                         double* virial,
                         size_t virial_pitch);

/* ---- integrator ---- */
void orc_sum_net_force(orc_d4* net_force,
                       double* net_virial,
                       size_t net_virial_pitch,
                       const orc_d4* const* forces,
                       const double* const* virials,
                       const size_t* virial_pitches,
                       unsigned int n_forces,
                       unsigned int n_particles);
void orc_compute_accel(const orc_d4* vel, const orc_d4* net_force, orc_d3* accel, unsigned int N);
void orc_step_one(orc_d4* pos,
                  orc_d4* vel,
                  const orc_d3* accel,
                  orc_i3* image,
                  const unsigned int* group,
                  unsigned int group_size,
                  const orc_box_t* box,
                  double rescale,
                  double dt);
void orc_step_two(orc_d4* vel,
                  orc_d3* accel,
                  const unsigned int* group,
                  unsigned int group_size,
                  const orc_d4* net_force,
                  double rescale,
                  double dt);
/* hoomd/md/ComputeThermo.cc:99-150 : kinetic energy 1/2 m v^2 summed, potential = sum force.w */
void orc_thermo(const orc_d4* vel, const orc_d4* net_force, unsigned int N, double* ke, double* pe);

/* ---- whole-path driver used for NVE drift and the CPU baseline ---- */
typedef struct orc_sim orc_sim_t;
orc_sim_t* orc_sim_create(unsigned int N,
                          const orc_d4* pos,
                          const orc_d4* vel,
                          const double Lbox[3],
                          double r_cut,
                          double r_buff,
                          double epsilon,
                          double sigma,
                          double dt,
                          int shift_mode);
void orc_sim_destroy(orc_sim_t* s);
/* runs prepRun if needed, then nsteps of IntegratorTwoStep::update */
void orc_sim_run(orc_sim_t* s, uint64_t nsteps);
void orc_sim_get(orc_sim_t* s, orc_d4* pos, orc_d4* vel, orc_i3* image, orc_d4* net_force);
void orc_sim_energies(orc_sim_t* s, double* ke, double* pe);
void orc_sim_stats(orc_sim_t* s, uint64_t* num_builds, unsigned int* shortest_rebuild, uint64_t* timestep);

#ifdef __cplusplus
}
#endif
#endif
