/* hoomd_amd.h -- C ABI of the MI355X (gfx950) kernel-driver layer for HOOMD-blue's short-range MD step.
 *
 * Each entry point replaces one of the reference's GPU kernel drivers (free functions in
 * namespace hoomd[::md]::kernel that take raw device pointers); the reference interface it
 * stands in for is cited as file:line relative to the HOOMD-blue 4.1.0 source tree.
 *
 * Conventions
 *   - every pointer named d_* is device memory owned by the caller; drivers never allocate;
 *   - Scalar is double: hmd_scalar4 == Scalar4 (32 B), hmd_scalar3 == Scalar3 (24 B);
 *   - the particle type id is bit-stuffed in the low 32 bits of pos.w (hoomd/HOOMDMath.h:166-217),
 *     mass travels in vel.w, potential energy in force.w;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream the reference uses);
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     HMD_ERR_* code when the arguments are rejected before anything is launched;
 *   - launches are asynchronous; overflow / NaN / out-of-box conditions are reported through
 *     the same small device flag arrays the reference uses and are read by the caller.
 */
#ifndef HOOMD_AMD_H
#define HOOMD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMD_ABI_VERSION 1

#define HMD_OK 0
#define HMD_ERR_INVALID_ARGUMENT (-1) /* NULL where a pointer is required, zero-sized indexer ... */
#define HMD_ERR_UNSUPPORTED (-2)      /* valid in the reference, not built here (see message)    */
#define HMD_ERR_TOO_LARGE (-3)        /* exceeds a documented fixed limit                         */

#define HMD_NO_BODY 0xffffffffu

typedef struct
    {
    double x, y, z, w;
    } hmd_scalar4;
typedef struct
    {
    double x, y, z;
    } hmd_scalar3;
typedef struct
    {
    int x, y, z;
    } hmd_int3;
typedef struct
    {
    unsigned int x, y;
    } hmd_uint2;
typedef struct
    {
    unsigned int x, y, z;
    } hmd_uint3;
typedef struct
    {
    unsigned int x, y, z, w;
    } hmd_uint4;

/* BoxDim (hoomd/BoxDim.h:633-641) as plain data.  nearest_plane holds
   BoxDim::getNearestPlaneDistance() (hoomd/BoxDim.h:536-552) evaluated once on the host so that
   host and device bin with the identical value. */
typedef struct
    {
    double lo[3];
    double hi[3];
    double L[3];
    double Linv[3];
    double xy, xz, yz;
    double nearest_plane[3];
    unsigned char periodic[3];
    unsigned char pad_[5];
    } hmd_box_t;

/* Index2D / Index3D (hoomd/Index1D.h:32-89, :95-165): (i,j) -> j*w+i ; (i,j,k) -> (k*h+j)*w+i */
typedef struct
    {
    unsigned int w, h;
    } hmd_index2d_t;
typedef struct
    {
    unsigned int w, h, d;
    } hmd_index3d_t;

/* Fills every field of *box from lengths + tilt factors (hoomd/BoxDim.h:125-130,174-187). Host only. */
void hmd_box_init(hmd_box_t* box, double Lx, double Ly, double Lz, double xy, double xz, double yz);
/* As BoxDim::setLoHi (hoomd/BoxDim.h:209-221); tilt factors are kept. Host only. */
void hmd_box_set_lo_hi(hmd_box_t* box, const double lo[3], const double hi[3]);

int hmd_abi_version(void);
/* Text for the last negative return code on this host thread. */
const char* hmd_last_error(void);

/* --------------------------------------------------------------------------------------------
 * Cell list.  Replaces gpu_compute_cell_list (hoomd/CellListGPU.cuh:23-44, hoomd/CellListGPU.cu:173-244).
 * The caller zeroes d_cell_size first (hoomd/CellListGPU.cc:96-99) unless clear_cell_size != 0.
 * d_xyzf / d_type_body / d_cell_idx / d_cell_orientation may be NULL to skip that output.
 * xyzf.w receives the charge (flag_charge), the type (flag_type) or else the particle index.
 * d_conditions: x = 1 + largest overflowing slot, y = 1 + index of a NaN particle,
 *               z = 1 + index of an owned particle outside the box (hoomd/CellList.cc:549-607).
 * ------------------------------------------------------------------------------------------ */
typedef struct
    {
    unsigned int* d_cell_size;
    hmd_scalar4* d_xyzf;
    hmd_uint2* d_type_body;
    hmd_scalar4* d_cell_orientation;
    unsigned int* d_cell_idx;
    hmd_uint3* d_conditions;
    const hmd_scalar4* d_pos;
    const hmd_scalar4* d_orientation;
    const double* d_charge;
    const double* d_diameter;
    const unsigned int* d_body;
    unsigned int N;
    unsigned int n_ghost;
    unsigned int Nmax;
    int flag_charge;
    int flag_type;
    hmd_box_t box;
    hmd_index3d_t ci;
    hmd_index2d_t cli;
    hmd_scalar3 ghost_width;
    unsigned int block_size; /* 0 = library default */
    int clear_cell_size;     /* non-zero: zero d_cell_size on `stream` before binning */
    } hmd_cell_list_args_t;
int hmd_compute_cell_list(const hmd_cell_list_args_t* args, void* stream);

/* Deterministic mode: orders the members of every cell by particle index.
   Replaces gpu_sort_cell_list (hoomd/CellListGPU.cuh:64-76, hoomd/CellListGPU.cu:478-565);
   done per cell in LDS instead of a global key sort, so no temporaries are needed.
   Cells holding more than 1024 members are rejected with HMD_ERR_TOO_LARGE. */
typedef struct
    {
    const unsigned int* d_cell_size;
    hmd_scalar4* d_xyzf;
    hmd_uint2* d_type_body;
    unsigned int* d_cell_idx; /* optional */
    hmd_index3d_t ci;
    hmd_index2d_t cli;
    int flag_is_index; /* xyzf.w holds the particle index (sort key); otherwise d_cell_idx is required */
    } hmd_sort_cell_list_args_t;
int hmd_sort_cell_list(const hmd_sort_cell_list_args_t* args, void* stream);

/* --------------------------------------------------------------------------------------------
 * Neighbor list.
 * ------------------------------------------------------------------------------------------ */

/* Replaces gpu_nlist_needs_update_check_new (hoomd/md/NeighborListGPU.cu:86-128): sets
   *d_result = checkn (atomic max) when any of the N particles moved at least
   ((rcut_max[type]+r_buff)*lambda_min - rcut_max[type])/2 from lambda*last_pos (minimum image). */
typedef struct
    {
    unsigned int* d_result;
    const hmd_scalar4* d_last_pos;
    const hmd_scalar4* d_pos;
    unsigned int N;
    hmd_box_t box;
    const double* d_rcut_max;
    double r_buff;
    unsigned int ntypes;
    double lambda_min;
    hmd_scalar3 lambda;
    unsigned int checkn;
    /* Optional low-latency readback.  When d_published is not NULL (a device-visible address of
       pinned host memory, hipHostGetDevicePointer), a one-lane kernel queued behind the check
       stores ((uint64_t)stamp << 32) | *d_result there with a single system-scope 8-byte store;
       the host can then poll that word instead of synchronising the stream (the reference reads
       the flag back with a blocking copy, hoomd/md/NeighborListGPU.cc:77-82). */
    unsigned long long* d_published;
    unsigned int stamp;
    } hmd_nlist_check_args_t;
int hmd_nlist_needs_update_check(const hmd_nlist_check_args_t* args, void* stream);

/* The low-latency readback on its own: queues a one-wave kernel that stores (stamp << 32 | *d_word) to
   d_published (see hmd_nlist_check_args_t).  For a flag that changes after the check, e.g. once it has
   been reduced over the ranks of a domain decomposition (the role of the MPI_Allreduce in
   hoomd/md/NeighborListGPU.cc:86-98).  No counterpart kernel in the reference. */
int hmd_publish_word(const unsigned int* d_word, unsigned long long* d_published, unsigned int stamp, void* stream);
/* The same for n <= 64 consecutive words: d_published[k] = (stamp << 32 | d_words[k]).  Used for the overflow
   conditions of the cell list and the neighbor list, which the reference reads back with a blocking copy
   (hoomd/CellListGPU.cc:150-170, hoomd/md/NeighborListGPU.cc checkConditions). */
/* Does `stream` (a created stream of the current GPU) run beside the default stream, or behind it?  HIP maps its
   streams onto a few hardware queues, and one that shares the default stream's queue cannot overlap with it.  Runs a
   120 us probe (a long kernel on the default stream, a one-wave kernel on `stream`) and sets *overlaps to 1 when the
   small kernel finished in the first half of the long one.  d_scratch2: two 64-bit device words.  Synchronises the
   device; for set-up time.  No counterpart in the reference (its communicator finishes the ghost update before
   the forces start, hoomd/Communicator.cc:1544-1549); used to choose the stream of the halo exchange. */
int hmd_stream_runs_beside_default(void* stream, unsigned long long* d_scratch2, int* overlaps);
int hmd_publish_words(const unsigned int* d_words, unsigned int n, unsigned long long* d_published, unsigned int stamp, void* stream);

/* Replaces gpu_nlist_build_head_list (hoomd/md/NeighborListGPU.cu:415-456):
   d_head_list[i] = sum_{j<i} d_Nmax[type_j];  *d_req_size_nlist = sum over all N.
   d_workspace must hold hmd_nlist_head_list_workspace_bytes(N) bytes. */
size_t hmd_nlist_head_list_workspace_bytes(unsigned int N);
typedef struct
    {
    size_t* d_head_list;
    size_t* d_req_size_nlist;
    const unsigned int* d_Nmax;
    const hmd_scalar4* d_pos;
    unsigned int N;
    unsigned int ntypes;
    void* d_workspace;
    size_t workspace_bytes;
    } hmd_head_list_args_t;
int hmd_nlist_build_head_list(const hmd_head_list_args_t* args, void* stream);

/* hmd_nlist_build_head_list without any allocation, for callers that cannot allocate device memory or ask
   their caller for a workspace: code under HIP graph capture, tight serving loops, harnesses that forbid
   hipMalloc.  No counterpart in the reference.  d_workspace / workspace_bytes are ignored: the tile sums of
   the scan live in a static device array inside the library (64 KiB per GPU, held until the process
   ends), sized for N <= 16,777,216; a larger N returns HMD_ERR_TOO_LARGE.  Same outputs as
   hmd_nlist_build_head_list.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU
   (different streams or host threads) race and must be serialised by the caller.  The array used is the
   copy of the current GPU; a non-null stream must belong to that GPU (HMD_ERR_INVALID_ARGUMENT otherwise).
   The FIRST call on a GPU looks the array's address up (a lock and one hipGetSymbolAddress): make it
   outside graph capture; later calls only launch kernels. */
int hmd_nlist_build_head_list_noalloc(const hmd_head_list_args_t* args, void* stream);

/* Replaces gpu_compute_nlist_binned (hoomd/md/NeighborListGPUBinned.cuh:35-62,
   hoomd/md/NeighborListGPUBinned.cu:756-823).  Full neighbor list only (as the reference's GPU
   class, hoomd/md/NeighborListGPUBinned.cc:51-54).  For every owned particle i: all j != i in the
   adjacent cells with r_cut(ti,tj) > 0 and |minImage(ri-rj)|^2 <= (r_cut(ti,tj)+r_buff)^2, in
   unspecified order, at d_nlist[d_head_list[i] + k]; d_n_neigh[i] counts all hits, even those
   beyond d_Nmax[ti], and then d_conditions[ti] = max(.., n_neigh) signals the overflow.
   d_last_updated_pos[i] = pos[i].  d_cell_xyzf.w must hold particle indices and
   d_cell_type_body the (type, body) pairs written by hmd_compute_cell_list.
   threads_per_particle and block_size are accepted for interface parity; 0 = library default. */
typedef struct
    {
    unsigned int* d_nlist;
    unsigned int* d_n_neigh;
    hmd_scalar4* d_last_updated_pos;
    unsigned int* d_conditions;
    const unsigned int* d_Nmax;
    const size_t* d_head_list;
    const hmd_scalar4* d_pos;
    const unsigned int* d_body;
    unsigned int N;
    const unsigned int* d_cell_size;
    const hmd_scalar4* d_cell_xyzf;
    const unsigned int* d_cell_idx; /* unused: xyzf carries the index */
    const hmd_uint2* d_cell_type_body;
    const unsigned int* d_cell_adj;
    hmd_index3d_t ci;
    hmd_index2d_t cli;
    hmd_index2d_t cadji;
    hmd_box_t box;
    const double* d_r_cut;
    double r_buff;
    unsigned int ntypes;
    unsigned int threads_per_particle;
    unsigned int block_size;
    int filter_body;
    hmd_scalar3 ghost_width;
    /* Optional (NULL: none): exclusions by particle index, the arrays hmd_update_exclusion_list fills.  Excluded
       partners are left out while the rows are built, which gives the rows and counts that a build followed by
       hmd_nlist_filter gives (hoomd/md/NeighborListGPU.cu:158-223), without the second pass over the rows. */
    const unsigned int* d_n_ex_idx;
    const unsigned int* d_ex_list_idx; /* entry o of particle i at exli(i, o) */
    hmd_index2d_t exli;
    } hmd_nlist_binned_args_t;
int hmd_compute_nlist_binned(const hmd_nlist_binned_args_t* args, void* stream);

/* Replaces gpu_nlist_filter (hoomd/md/NeighborListGPU.cu:225-267): removes from every row the
   entries found in that particle's exclusion list d_ex_list[exli(idx, k)], k < d_n_ex[idx]. */
typedef struct
    {
    unsigned int* d_n_neigh;
    unsigned int* d_nlist;
    const size_t* d_head_list;
    const unsigned int* d_n_ex;
    const unsigned int* d_ex_list;
    hmd_index2d_t exli;
    unsigned int N;
    unsigned int block_size;
    } hmd_nlist_filter_args_t;
int hmd_nlist_filter(const hmd_nlist_filter_args_t* args, void* stream);

/* Replaces gpu_update_exclusion_list (hoomd/md/NeighborListGPU.cu:312-340). */
typedef struct
    {
    const unsigned int* d_tag;
    const unsigned int* d_rtag;
    const unsigned int* d_n_ex_tag;
    const unsigned int* d_ex_list_tag;
    hmd_index2d_t ex_list_tag_indexer;
    unsigned int* d_n_ex_idx;
    unsigned int* d_ex_list_idx;
    hmd_index2d_t ex_list_indexer;
    unsigned int N;
    } hmd_update_exclusion_args_t;
int hmd_update_exclusion_list(const hmd_update_exclusion_args_t* args, void* stream);

/* --------------------------------------------------------------------------------------------
 * Pair forces.  Replaces gpu_compute_pair_forces<evaluator>(const pair_args_t&, const param_type*)
 * (hoomd/md/PotentialPairGPU.cuh:44-98 for the argument block, :579-656 for the driver); one
 * exported symbol per evaluator, as the reference instantiates one driver per evaluator
 * (hoomd/md/CMakeLists.txt:498-547, example_plugins/pair_plugin/PotentialPairExampleGPUKernel.cu:13-15).
 * shift_mode: 0 none, 1 shift, 2 xplor (hoomd/md/PotentialPair.h:126-131).
 * d_force[i] = (F_i, sum_j E_ij/2); d_virial[c*virial_pitch+i], c = xx,xy,xz,yy,yz,zz, only when
 * compute_virial != 0.  Rows of owned particles only; ghost indices may appear as neighbors.
 * ------------------------------------------------------------------------------------------ */
typedef struct
    {
    hmd_scalar4* d_force;
    double* d_virial;
    size_t virial_pitch;
    unsigned int N;
    unsigned int n_max;
    const hmd_scalar4* d_pos;
    const double* d_charge;
    hmd_box_t box;
    const unsigned int* d_n_neigh;
    const unsigned int* d_nlist;
    const size_t* d_head_list;
    const double* d_rcutsq;
    const double* d_ronsq;
    size_t size_neigh_list;
    unsigned int ntypes;
    unsigned int block_size;           /* 0 = library default */
    unsigned int shift_mode;
    unsigned int compute_virial;
    unsigned int threads_per_particle; /* power of two in [1,64]; 0 = library default */
    /* Optional: when d_skip_flag is not NULL and *d_skip_flag == skip_value at launch, the kernel does
       nothing.  Lets a caller queue the launch behind the neighbor-list distance check
       (hmd_nlist_needs_update_check: d_result / checkn) before the host has read the check's result:
       if the list turns out stale the launch is a no-op and the caller launches again after the
       rebuild. */
    const unsigned int* d_skip_flag;
    unsigned int skip_value;
    /* Optional, with d_skip_flag: the kernel also hands the word to the host, storing
       (publish_stamp << 32 | *d_skip_flag) to d_published as hmd_publish_word would, before anything else. */
    unsigned int publish_stamp;
    unsigned long long* d_published;
    /* Tile entry points only (ignored by the row-based ones): which blocks of 2x2x2 cells this launch
       serves.  0: all; 1: the blocks b with block_lo[d] <= b[d] <= block_hi[d] on every axis d;
       2: all blocks outside that window.  Under domain decomposition a caller computes the window of
       blocks whose 4x4x4 regions hold no ghosts while the ghost positions are still in flight (mode 1,
       on its own stream) and the rest once they have arrived (mode 2).  The reference has no such
       split: it waits for the ghosts, then launches over all particles (hoomd/md/PotentialPairGPU.h:89-160). */
    unsigned int block_window_mode;
    unsigned int block_lo[3];
    unsigned int block_hi[3];
    } hmd_pair_args_t;

/* EvaluatorPairLJ::param_type (hoomd/md/EvaluatorPairLJ.h:99-155) */
typedef struct
    {
    double sigma_6;
    double epsilon_x_4;
    } hmd_lj_params_t;
/* EvaluatorPairExample::param_type (example_plugins/pair_plugin/EvaluatorPairExample.h:37-76) */
typedef struct
    {
    double k;
    double sigma;
    } hmd_example_params_t;

int hmd_compute_pair_forces_lj(const hmd_pair_args_t* args, const hmd_lj_params_t* d_params, void* stream);
int hmd_compute_pair_forces_example(const hmd_pair_args_t* args, const hmd_example_params_t* d_params, void* stream);

/* --------------------------------------------------------------------------------------------
 * Bonds (Kremer-Grest chains: FENE + WCA).
 * ------------------------------------------------------------------------------------------ */

/* fene_params (hoomd/md/EvaluatorBondFENE.h:30-69) */
typedef struct
    {
    double k;
    double r_0;
    double epsilon_x_4;
    double sigma_6;
    double delta;
    } hmd_fene_params_t;

/* Replaces gpu_update_group_table<2> (hoomd/BondedGroupData.cu:125-220): the per-particle bond table
   in the current particle order.  Input is the same table keyed by tag, which only changes when
   bonds are added or removed: bond k of the particle with tag t is d_tag_blist[tag_blist_idx(t, k)] =
   (partner tag, bond type), k < d_tag_n_bonds[t], in the order the bonds were defined.  Output, for
   the particle at index i: d_n_bonds[i] and d_blist[blist_idx(i, k)] = (partner index, bond type)
   (hoomd/BondedGroupData.h: gpu table, group_storage<2>).  Indexers are (pitch, max bonds) with the
   particle/tag coordinate fastest.  A partner that is not local (rtag >= n_total) sets *d_flags = 1
   (the reference's "bond incomplete" error, hoomd/md/PotentialBond.h:228-234). */
typedef struct
    {
    unsigned int N;       /* local particles */
    unsigned int n_total; /* N + ghosts */
    const unsigned int* d_tag;
    const unsigned int* d_rtag;
    const unsigned int* d_tag_n_bonds;
    const hmd_uint2* d_tag_blist;
    hmd_index2d_t tag_blist_idx;
    unsigned int* d_n_bonds;
    hmd_uint2* d_blist;
    hmd_index2d_t blist_idx;
    unsigned int* d_flags;
    } hmd_bond_table_args_t;
int hmd_update_bond_table(const hmd_bond_table_args_t* args, void* stream);

/* Replaces gpu_compute_bond_forces<EvaluatorBondFENE, 2> (hoomd/md/PotentialBondGPU.cuh:97-239, driver
   :241-330): one thread per particle walks its bond table; each end of a bond gets the force and
   half the bond energy.  d_virial may be NULL with compute_virial == 0.  A bond stretched beyond
   r_0 sets *d_flags = 1 and leaves that particle's outputs unwritten (the reference raises "bond out
   of bounds" from the flag, hoomd/md/PotentialBondGPU.h:130-140). */
typedef struct
    {
    hmd_scalar4* d_force;
    double* d_virial;
    size_t virial_pitch;
    unsigned int N;
    const hmd_scalar4* d_pos;
    hmd_box_t box;
    const hmd_uint2* d_blist;
    hmd_index2d_t blist_idx;
    const unsigned int* d_n_bonds;
    unsigned int n_bond_types;
    unsigned int* d_flags;
    int compute_virial;
    } hmd_bond_args_t;
int hmd_compute_bond_forces_fene(const hmd_bond_args_t* args, const hmd_fene_params_t* d_params, void* stream);

/* --------------------------------------------------------------------------------------------
 * Integrator.
 * ------------------------------------------------------------------------------------------ */

/* Replaces gpu_integrator_sum_net_force (hoomd/Integrator.cu:171-230, force list struct
   hoomd/Integrator.cuh:23-77): net = (clear ? 0 : net) + sum of up to 6 force computes over
   nparticles (owned + ghosts).  Torque pointers (net and per force) may be NULL: point particles. */
typedef struct
    {
    const hmd_scalar4* f[6];
    const hmd_scalar4* t[6];
    const double* v[6];
    size_t vpitch[6];
    } hmd_force_list_t;
typedef struct
    {
    hmd_scalar4* d_net_force;
    double* d_net_virial;
    size_t net_virial_pitch;
    hmd_scalar4* d_net_torque;
    hmd_force_list_t force_list;
    unsigned int nparticles;
    int clear;
    int compute_virial;
    } hmd_sum_net_force_args_t;
int hmd_integrator_sum_net_force(const hmd_sum_net_force_args_t* args, void* stream);

/* Replaces gpu_nvt_rescale_step_one (hoomd/md/TwoStepConstantVolumeGPU.cu:102-155):
   v += a*dt/2 ; v *= rescale_factor ; x += v*dt ; wrap(x, image), for the group members.
   d_group_members may be NULL for the identity group 0..group_size-1.
   The displacement limit follows the CPU class (hoomd/md/TwoStepConstantVolume.cc:45-53). */
typedef struct
    {
    hmd_scalar4* d_pos;
    hmd_scalar4* d_vel;
    const hmd_scalar3* d_accel;
    hmd_int3* d_image;
    const unsigned int* d_group_members;
    unsigned int group_size;
    hmd_box_t box;
    unsigned int block_size;
    double rescale_factor;
    double deltaT;
    int use_limit;
    double maximum_displacement;
    /* Optional: the factor is sqrt(rescale_gamma / *d_rescale_ke), evaluated on the device, and rescale_factor is
       ignored.  For a thermostat whose factor is a random target energy over the kinetic energy of this very time step
       (BussiThermostat, hoomd/md/Thermostat.h:348-371): the kinetic energy stays where hmd_compute_thermo left it
       (d_out[0]) and the host does not wait for it.  Division and square root are the correctly rounded ones, so the
       factor has the bits the host would compute from a copy of the same sum. */
    const double* d_rescale_ke;
    double rescale_gamma;
    } hmd_step_one_args_t;
int hmd_nvt_rescale_step_one(const hmd_step_one_args_t* args, void* stream);

/* Replaces gpu_nvt_rescale_step_two (hoomd/md/TwoStepConstantVolumeGPU.cu:215-260):
   a = F * (1/m) ; v *= rescale_factor ; v += a*dt/2 (CPU-class arithmetic,
   hoomd/md/TwoStepConstantVolume.cc:243-252). */
typedef struct
    {
    hmd_scalar4* d_vel;
    hmd_scalar3* d_accel;
    const unsigned int* d_group_members;
    unsigned int group_size;
    const hmd_scalar4* d_net_force;
    unsigned int block_size;
    double deltaT;
    double rescale_factor;
    /* Optional: a second force array.  The net force is then d_net_force + d_net_force2, as
       hmd_integrator_sum_net_force would have left it for two forces (hoomd/Integrator.cu:67-166), added where it is
       read: a run with two forces and no virial request needs no pass of its own for the sum.  Also honoured by
       hmd_nve_step_two_one_check (two.d_net_force2). */
    const hmd_scalar4* d_net_force2;
    } hmd_step_two_args_t;
int hmd_nvt_rescale_step_two(const hmd_step_two_args_t* args, void* stream);

/* Second half of step t, first half of step t+1 and the neighbor-list distance check of the new
   positions in ONE pass over the particles: the three reference drivers gpu_nvt_rescale_step_two
   (hoomd/md/TwoStepConstantVolumeGPU.cu:215-260), gpu_nvt_rescale_step_one (:102-155) and
   gpu_nlist_needs_update_check_new (hoomd/md/NeighborListGPU.cu:86-97) run back to back on the
   same particles when nothing else happens between two time steps; results are bit-identical to
   calling them in that order.  `two` and `one` must name the same group.  check.d_result == NULL
   skips the check; otherwise the group must be every particle (d_group_members NULL,
   group_size == check.N).  With check.d_published set, (stamp << 32 | *d_result) is published by a
   one-wave kernel queued behind the pass.  With d_cpos / d_slot_of set, the new position of particle i
   is also stored at d_cpos[d_slot_of[i]] (what hmd_tile_refresh would do afterwards).
   two.d_accel may be NULL (one.d_accel is never read: the acceleration passes from the second-half
   part to the first-half part in registers): the accelerations of time t are then not stored, and a
   caller that needs the array runs hmd_nvt_rescale_step_two on its own for the last step. */
typedef struct
    {
    hmd_step_two_args_t two;
    hmd_step_one_args_t one;
    hmd_nlist_check_args_t check;
    const unsigned int* d_slot_of;
    hmd_scalar4* d_cpos;
    } hmd_fused_step_args_t;
int hmd_nve_step_two_one_check(const hmd_fused_step_args_t* args, void* stream);
/* The first half step alone with the same riders, for integration methods whose half steps cannot be joined (a
   thermostat needs the kinetic energy of the finished step before it can scale the next one): gpu_nvt_rescale_step_one
   (hoomd/md/TwoStepConstantVolumeGPU.cu:102-155) + gpu_nlist_needs_update_check_new (hoomd/md/NeighborListGPU.cu:86-97)
   + the slot-ordered position copy, bit-identical to calling them in that order.  `two` is ignored; one.d_accel is
   read; check / d_cpos / d_slot_of as above. */
int hmd_nvt_rescale_step_one_check(const hmd_fused_step_args_t* args, void* stream);

/* a = F/m for all N particles: Integrator::computeAccelerations (hoomd/Integrator.cc:106-127),
   which the reference runs on the host; here it stays on the device. */
int hmd_compute_accelerations(hmd_scalar3* d_accel,
                              const hmd_scalar4* d_vel,
                              const hmd_scalar4* d_net_force,
                              unsigned int N,
                              void* stream);

/* Kinetic energy, potential energy and the virial trace of a group, reduced on the device.
   Replaces gpu_compute_thermo_partial + gpu_compute_thermo_final
   (hoomd/md/ComputeThermoGPU.cu:46-130,326-520).  d_out[0] = sum m v^2 / 2, d_out[1] = sum
   net_force.w, d_out[2] = sum (virial xx+yy+zz) (0 when d_net_virial is NULL), d_out[3..5] = total
   momentum.  d_scratch needs hmd_thermo_scratch_bytes() bytes; d_out 8 doubles. */
size_t hmd_thermo_scratch_bytes(void);
typedef struct
    {
    double* d_out;
    void* d_scratch;
    const hmd_scalar4* d_vel;
    const hmd_scalar4* d_net_force;
    const double* d_net_virial;
    size_t net_virial_pitch;
    const unsigned int* d_group_members; /* NULL = identity */
    unsigned int group_size;
    /* optional: the device's view of eight doubles of mapped pinned host memory that receive the same values (a host
       that does not wait for the sums can still look at the latest finished ones) */
    double* d_out_host;
    } hmd_thermo_args_t;
int hmd_compute_thermo(const hmd_thermo_args_t* args, void* stream);
/* gpu_nvt_rescale_step_two (hoomd/md/TwoStepConstantVolumeGPU.cu:215-260) followed by the thermo sums of the new
   velocities (hoomd/md/ComputeThermoGPU.cu:46-130,326-520) in ONE pass over the particles: with a thermostat the sums
   of the finished step are needed before the next first half step, and the velocities are in registers here.
   Bit-identical to hmd_nvt_rescale_step_two followed by hmd_compute_thermo (same tree of partial sums).  `thermo`
   must name the same group, velocities and net force as `two`, without virial (d_net_virial NULL) and without a
   second force array; HMD_ERR_UNSUPPORTED otherwise. */
int hmd_nvt_rescale_step_two_thermo(const hmd_step_two_args_t* two, const hmd_thermo_args_t* thermo, void* stream);

/* --------------------------------------------------------------------------------------------
 * Particle reordering by cell (data locality for the neighbor gathers).  Stands in for the
 * reference's space-filling-curve sorter kernels gpu_sfc_bin_particles / gpu_apply_sorted_order
 * (hoomd/SFCPackTunerGPU.cu:26-160): d_order[k] = old index of the particle that becomes index k,
 * taken from a cell list (cells walked in memory order, members in slot order).
 * ------------------------------------------------------------------------------------------ */
typedef struct
    {
    unsigned int* d_order;         /* out: N entries */
    unsigned int* d_cell_offset;   /* scratch: ncells+1 entries */
    const unsigned int* d_cell_size;
    const hmd_scalar4* d_cell_xyzf; /* w = particle index */
    hmd_index3d_t ci;
    hmd_index2d_t cli;
    unsigned int N;                /* owned particles; entries with index >= N are skipped */
    void* d_workspace;
    size_t workspace_bytes;        /* >= hmd_cell_order_workspace_bytes(ncells) */
    } hmd_cell_order_args_t;
size_t hmd_cell_order_workspace_bytes(unsigned int ncells);
int hmd_cell_order(const hmd_cell_order_args_t* args, void* stream);

/* out[k] = in[d_order[k]] for arrays of elem_bytes-sized elements (4, 8, 12, 16, 24 or 32). */
int hmd_gather_rows(void* d_out, const void* d_in, const unsigned int* d_order, unsigned int N,
                    unsigned int elem_bytes, void* stream);
/* d_rtag[d_tag[k]] = k */
int hmd_build_rtag(unsigned int* d_rtag, const unsigned int* d_tag, unsigned int N, void* stream);

/* --------------------------------------------------------------------------------------------
 * Halo exchange helpers (the CommunicatorGPU role, hoomd/CommunicatorGPU.cu).
 * ------------------------------------------------------------------------------------------ */

/* Ghost plan: bit d of d_plan[i] is set when owned particle i lies within r_ghost of face d of
   the local box (0:+x 1:-x 2:+y 3:-y 4:+z 5:-z), only in decomposed directions.
   Replaces gpu_make_ghost_exchange_plan (hoomd/CommunicatorGPU.cu:306-372). */
typedef struct
    {
    unsigned int* d_plan;
    const hmd_scalar4* d_pos;
    unsigned int N;
    hmd_box_t box;             /* local box */
    const double* d_r_ghost;   /* per type */
    unsigned int ntypes;
    unsigned int mask;         /* bit d set: direction d has a neighbouring domain */
    } hmd_ghost_plan_args_t;
int hmd_make_ghost_exchange_plan(const hmd_ghost_plan_args_t* args, void* stream);

/* Plan and send list in one pass: for every owned particle and every entry k of d_dir_mask (a
   combination of the plan bits above: one direction towards a neighboring domain, ghost_plan.mask limits
   the bits as before) whose bits are all set in the particle's plan, the key (k << 32 | particle index) is
   appended to d_keys and d_dir_count[k] is incremented; *d_total counts all keys, also those that did
   not fit `capacity` (the caller grows the buffer and runs again).  Keys arrive in no particular order:
   sorted, they are the send list ordered by direction, then particle index.  d_plan of ghost_plan may
   be NULL.  The caller zeroes d_total and d_dir_count.  Covers gpu_make_ghost_exchange_plan plus the
   per-direction selection and count of gpu_exchange_ghosts_make_indices (hoomd/CommunicatorGPU.cu:374-520). */
typedef struct
    {
    hmd_ghost_plan_args_t ghost_plan;
    const unsigned int* d_dir_mask;
    unsigned int n_dir;            /* <= 26 */
    unsigned long long* d_keys;
    unsigned int capacity;
    unsigned int* d_total;
    unsigned int* d_dir_count;
    } hmd_ghost_send_list_args_t;
int hmd_ghost_send_list(const hmd_ghost_send_list_args_t* args, void* stream);

/* out[k] = pos[d_send_idx[k]] + shift[d_send_shift_id[k]] (shift table of 27 periodic image
   vectors; id 13 = no shift).  Replaces gpu_pack_wrap_kernel / gpu_pack_kernel
   (hoomd/CommunicatorGPU.cu:778-875) for the position field. */
typedef struct
    {
    hmd_scalar4* d_out;
    const hmd_scalar4* d_pos;
    const unsigned int* d_send_idx;
    const unsigned char* d_send_shift_id;
    const hmd_scalar3* d_shift_table; /* 27 entries */
    unsigned int n_send;
    } hmd_pack_pos_args_t;
int hmd_pack_ghost_positions(const hmd_pack_pos_args_t* args, void* stream);

/* The same pack with the rebuild vote riding along: entry k goes to row d_out_row[k] of d_out, and row
   d_flag_rows[j] (one per destination rank, j < n_flag_rows) receives *d_flag_word in the low 32 bits of
   its first component.  hmd_unpack_ghost_positions is the receiving side: row r of d_in is copied to
   d_ghost_pos[d_map[r]] when d_map[r] >= 0, and is a vote row otherwise: *d_flag_word becomes the maximum
   of itself and the row's word.  When every rank exchanges ghosts with every other rank this replaces the
   separate reduction of the vote (MPI_Allreduce in hoomd/md/NeighborListGPU.cc:86-98) by the halo
   exchange that happens anyway.  No counterpart kernels in the reference. */
typedef struct
    {
    hmd_pack_pos_args_t pack;
    const unsigned int* d_out_row;
    const unsigned int* d_flag_word;
    const unsigned int* d_flag_rows;
    unsigned int n_flag_rows;
    } hmd_pack_vote_args_t;
int hmd_pack_ghost_positions_vote(const hmd_pack_vote_args_t* args, void* stream);
typedef struct
    {
    const hmd_scalar4* d_in;
    const int* d_map;
    unsigned int n_rows;
    hmd_scalar4* d_ghost_pos;
    unsigned int* d_flag_word;
    } hmd_unpack_vote_args_t;
int hmd_unpack_ghost_positions(const hmd_unpack_vote_args_t* args, void* stream);

/* Ascending sort of 64-bit keys, of which only the low key_bits bits are significant (the others must be equal in
   all keys).  The send lists of the exchange are built as unordered keys (destination << 32 | particle index);
   sorted they are grouped by destination in a reproducible order (the reference sorts with thrust behind its
   CachedAllocator, hoomd/CommunicatorGPU.cu:374-520).  d_keys_in and d_keys_out must differ. */
size_t hmd_sort_keys_workspace_bytes(unsigned int n);
int hmd_sort_keys(const unsigned long long* d_keys_in, unsigned long long* d_keys_out, unsigned int n, unsigned int key_bits,
                  void* d_workspace, size_t workspace_bytes, void* stream);

/* The per-particle arrays a migrating particle takes along (pdata_element, hoomd/ParticleData.h:153-175). */
typedef struct
    {
    hmd_scalar4* d_pos;
    hmd_scalar4* d_vel;
    hmd_scalar3* d_accel;
    hmd_int3* d_image;
    unsigned int* d_tag;
    unsigned int* d_body;
    double* d_charge;
    double* d_diameter;
    } hmd_pdata_arrays_t;
/* Split sorted send-list keys (direction << 32 | particle index, from hmd_ghost_send_list) into the two arrays
   the pack kernels read: d_send_idx[k] = index, d_send_dir[k] = direction (the selection output of
   gpu_exchange_ghosts_make_indices, hoomd/CommunicatorGPU.cu:374-520). */
int hmd_ghost_split_keys(const unsigned long long* d_keys, unsigned int n, unsigned int* d_send_idx, unsigned char* d_send_dir, void* stream);

/* First exchange after a new ghost plan: like hmd_pack_ghost_positions, and the particle's tag rides in the
   upper half of the w component (whose lower half is the type id), so one message carries what the reference
   sends as separate tag and position messages (hoomd/CommunicatorGPU.cu:778-875, hoomd/Communicator.cc
   exchangeGhosts).  hmd_unpack_ghost_init is the receiving side: ghost first + k gets position and type of
   row k, its tag, and the defaults of the fields that are not communicated (image 0, body none, charge 0,
   diameter 1). */
int hmd_pack_ghost_init(const hmd_pack_pos_args_t* args, const unsigned int* d_tag, void* stream);
typedef struct
    {
    const hmd_scalar4* d_in;
    unsigned int n;
    unsigned int first;
    hmd_pdata_arrays_t arrays; /* d_vel and d_accel are not touched and may be NULL */
    } hmd_unpack_ghost_init_args_t;
int hmd_unpack_ghost_init(const hmd_unpack_ghost_init_args_t* args, void* stream);

/* The same first exchange in messages whose sizes both ends know beforehand, so that no exchange of counts precedes
   it: the message to (from) rank r starts at row row0[r] of the buffer with a header row -- the number of data rows
   the sender had for r in the low word of x, the sender's `flag` in the low word of y -- followed by room for
   cap[r] data rows.  off: exclusive prefix sums of the data rows per rank (entries off[r] .. off[r + 1] - 1 of the
   send list go to rank r; ghosts first + off[r] .. come from rank r); rows that do not fit their message are left
   out, which the sender announces to every receiver through `flag`.  hmd_ghost_slot_headers copies the two words
   of every incoming header to d_words[2 r], d_words[2 r + 1].  (The reference sizes its receive buffers from an
   MPI exchange of counts before every ghost exchange, hoomd/CommunicatorGPU.cc:2396-2420; no counterpart.) */
#define HMD_GHOST_SLOTS_MAX_RANKS 64
typedef struct
    {
    unsigned int n_ranks;
    unsigned int flag;
    unsigned int off[HMD_GHOST_SLOTS_MAX_RANKS + 1];
    unsigned int row0[HMD_GHOST_SLOTS_MAX_RANKS];
    unsigned int cap[HMD_GHOST_SLOTS_MAX_RANKS];
    } hmd_ghost_slots_t;
int hmd_pack_ghost_init_slots(const hmd_pack_pos_args_t* args, const unsigned int* d_tag, const hmd_ghost_slots_t* slots, void* stream);
int hmd_ghost_slot_headers(const hmd_scalar4* d_in, const hmd_ghost_slots_t* slots, unsigned int* d_words, void* stream);
int hmd_unpack_ghost_init_slots(const hmd_unpack_ghost_init_args_t* args, const hmd_ghost_slots_t* slots, void* stream);

/* Words 0 and 1 of the header row of each of the n_ranks fixed-size migration messages in d_in (messages of
   slot_rows + 1 rows, see hmd_migrate_pack): d_words[2 r] = particles in the message from rank r,
   d_words[2 r + 1] = the largest count that rank sent to anyone.  One small kernel, so that the host reads the
   headers like its other counters instead of through a strided copy.  No counterpart in the reference. */
int hmd_migrate_slot_headers(const long long* d_in, unsigned int slot_rows, unsigned int n_ranks, unsigned int* d_words, void* stream);


/* Row maps for hmd_pack_ghost_positions_vote / hmd_unpack_ghost_positions from the message sizes: every
   message to or from another rank gets one extra row (the vote) after its data rows.  send_off / recv_off:
   exclusive prefix sums of the data rows per rank (n_ranks + 1 entries).  Outputs: d_out_row[send_off[n_ranks]],
   d_flag_rows[n_ranks - 1] (peers in rank order), d_in_map[recv_off[n_ranks] + n_ranks - 1]. */
#define HMD_VOTE_PLAN_MAX_RANKS 64
typedef struct
    {
    unsigned int n_ranks;
    unsigned int rank;
    unsigned int send_off[HMD_VOTE_PLAN_MAX_RANKS + 1];
    unsigned int recv_off[HMD_VOTE_PLAN_MAX_RANKS + 1];
    unsigned int* d_out_row;
    unsigned int* d_flag_rows;
    int* d_in_map;
    } hmd_vote_plan_args_t;
int hmd_vote_plan(const hmd_vote_plan_args_t* args, void* stream);

/* Owner rank of every owned particle after a move (the decision of gpu_stage_particles /
   gpu_select_particle_migrate, hoomd/CommunicatorGPU.cu:60-170, for a regular grid of domains with
   rank = (k * grid[1] + j) * grid[0] + i): fractional position in the global box, wrapped along
   decomposed directions, cut into grid cells.  d_dest[i] = owner; *d_n_leave counts particles whose
   owner is not `rank` (the caller zeroes it). */
typedef struct
    {
    const hmd_scalar4* d_pos;
    unsigned int N;
    double global_lo[3];
    double global_L[3];
    unsigned int grid[3];
    unsigned char decomposed[3];
    unsigned char pad_;
    unsigned int rank;
    int* d_dest;
    unsigned int* d_n_leave;
    /* Optional, all or none: the leavers as keys (owner << 32 | particle index) appended to d_keys in no
       particular order (sorted, they are the send list ordered by owner, then index; keys beyond `capacity`
       are dropped, *d_n_leave still counts them) and their number per owner in d_rank_count[n_ranks]
       (the caller zeroes it). */
    unsigned long long* d_keys;
    unsigned int capacity;
    unsigned int* d_rank_count;
    unsigned int n_ranks;
    } hmd_migrate_dest_args_t;
int hmd_migrate_dest(const hmd_migrate_dest_args_t* args, void* stream);

#define HMD_MIGRATE_ROW_WORDS 18 /* 64-bit words per particle in a migration message */

/* Pack the leavers into message rows (the role of gpu_pack / ParticleData::removeParticlesGPU's output,
   hoomd/ParticleData.cu:60-190, hoomd/CommunicatorGPU.cu:60-170): row k holds the particle of d_keys[k]
   (low 32 bits = index) as 18 words: pos.xyzw, vel.xyzw, accel.xyz (bit patterns of the doubles), image.xyz,
   tag, body (as signed 64-bit integers), charge, diameter (bit patterns).  The position is wrapped into the
   global box along decomposed directions and the image adjusted (hoomd/Communicator.cc:1189-1220). */
typedef struct
    {
    hmd_pdata_arrays_t arrays;
    const unsigned long long* d_keys;
    unsigned int n;
    double global_lo[3];
    double global_L[3];
    unsigned char decomposed[3];
    unsigned char pad_;
    long long* d_out;
    /* Optional fixed-size messages (slot_rows > 0, n_ranks <= 64): the message to rank r occupies rows
       [r * (slot_rows + 1), (r + 1) * (slot_rows + 1)) of d_out; its first row is a header (word 0: number of
       particles for r, word 1: the largest such number over all r, word 2: slot_rows), the particles follow;
       those beyond slot_rows are dropped (the receiver sees it in the header).  send_off: exclusive prefix sums of
       the keys per destination rank (n_ranks + 1 entries).  Receivers then know every message size without a
       separate exchange of counts.  With slot_rows == 0 rows are written back to back in key order. */
    unsigned int slot_rows;
    unsigned int n_ranks;
    unsigned int send_off[65];
    } hmd_migrate_pack_args_t;
int hmd_migrate_pack(const hmd_migrate_pack_args_t* args, void* stream);

/* The packs of the fixed-size messages with their sizes read ON THE DEVICE, where the kernels before them left them, so
   that the host does not fetch a count only to hand it back as an argument (the reference reads every count back and
   sizes its sends from it: hoomd/CommunicatorGPU.cc:1924-1942 migration, :2427-2446 ghosts).
   hmd_migrate_pack_slots_dev: hmd_migrate_pack in its fixed-size form (slot_rows >= 1; args->n and args->send_off are
   ignored) with n = min(*d_n_leave, capacity) keys of the sorted key array and the per-rank counts d_rank_count[r]
   (hmd_migrate_dest's outputs); places of the key array past the last key must hold a destination >= n_ranks.
   hmd_pack_ghost_init_slots_dev: hmd_pack_ghost_init_slots (slots->off and slots->flag are ignored) with
   n_send = min(*d_total, capacity) and the rows per destination rank summed from d_dir_count[k] over the directions k
   with d_dir_dest[k] == rank (hmd_ghost_send_list's outputs; directions sorted by destination rank).
   A list that outgrew its key array (*d_n_leave or *d_total above capacity) is incomplete; the sender says so in
   every header -- migration: 0xffffffff in word 1 (the "largest count"); ghosts: bit 1 of the flag word (bit 0: a
   message was too small) -- so that all ranks repeat the exchange together, the sender with a larger array.
   With d_sorted_keys (the sorted keys direction << 32 | particle index) hmd_pack_ghost_init_slots_dev takes the send
   list from the keys and stores it in args->d_send_idx / d_send_shift_id on the way (hmd_ghost_split_keys without a
   launch of its own).
   The *_and forms of the header kernels append n_extra (<= 64) words of d_extra behind the 2 * n_ranks header words,
   so that one look of the host covers the headers that arrived and the counts that stayed; with d_published they
   also publish every word, stamped, as hmd_publish_words would (one launch less before the host can look).
   hmd_comm_prep zeroes n_counters words and fills n_keys keys with all ones, the padding that sorts behind every key
   (the two memsets before hmd_migrate_dest / hmd_ghost_send_list in one launch). */
int hmd_comm_prep(unsigned int* d_counters, unsigned int n_counters, unsigned long long* d_keys, unsigned int n_keys, void* stream);
int hmd_migrate_pack_slots_dev(const hmd_migrate_pack_args_t* args, const unsigned int* d_n_leave, const unsigned int* d_rank_count,
                               unsigned int capacity, void* stream);
int hmd_migrate_slot_headers_and(const long long* d_in, unsigned int slot_rows, unsigned int n_ranks, const unsigned int* d_extra,
                                 unsigned int n_extra, unsigned int* d_words, unsigned long long* d_published, unsigned int stamp, void* stream);
int hmd_pack_ghost_init_slots_dev(const hmd_pack_pos_args_t* args, const unsigned int* d_tag, const hmd_ghost_slots_t* slots,
                                  const unsigned int* d_total, const unsigned int* d_dir_count, const unsigned int* d_dir_dest, unsigned int n_dir,
                                  unsigned int capacity, const unsigned long long* d_sorted_keys, void* stream);
int hmd_ghost_slot_headers_and(const hmd_scalar4* d_in, const hmd_ghost_slots_t* slots, const unsigned int* d_extra, unsigned int n_extra,
                               unsigned int* d_words, unsigned long long* d_published, unsigned int stamp, void* stream);

/* Close the gaps the leavers leave: the N - n_leave staying particles end up in [0, N - n_leave).  Stayers
   behind that boundary move into the leavers' places in front of it (in the order of d_keys, which must be
   sorted); everything else stays where it is.  d_dest: owner per particle as hmd_migrate_dest wrote it;
   d_scratch: 2 * n_leave words.  Stands in for the compaction of ParticleData::removeParticlesGPU
   (hoomd/ParticleData.cu:120-190), which keeps the order of the stayers instead. */
typedef struct
    {
    hmd_pdata_arrays_t arrays;
    const int* d_dest;
    unsigned int rank;
    unsigned int N;
    const unsigned long long* d_keys;
    unsigned int n_leave;
    unsigned int* d_scratch;
    } hmd_migrate_fill_args_t;
int hmd_migrate_fill(const hmd_migrate_fill_args_t* args, void* stream);

/* Append received message rows to the arrays: row k becomes particle first + k
   (ParticleData::addParticlesGPU, hoomd/ParticleData.cu:200-260). */
typedef struct
    {
    hmd_pdata_arrays_t arrays;
    const long long* d_in;
    unsigned int first;
    unsigned int n;
    /* Optional, the receiving side of fixed-size messages: row k of the particles comes from the message of the
       rank r with recv_off[r] <= k < recv_off[r + 1], at row r * (slot_rows + 1) + 1 + (k - recv_off[r]) of d_in. */
    unsigned int slot_rows;
    unsigned int n_ranks;
    unsigned int recv_off[65];
    } hmd_migrate_unpack_args_t;
int hmd_migrate_unpack(const hmd_migrate_unpack_args_t* args, void* stream);

/* --------------------------------------------------------------------------------------------
 * Tile path: the same neighbor list and pair forces, laid out for LDS instead of L1 gathers.
 * No single reference driver corresponds to these; together they stand in for
 * gpu_compute_nlist_binned (hoomd/md/NeighborListGPUBinned.cuh:35-62) followed by
 * gpu_compute_pair_forces<evaluator> (hoomd/md/PotentialPairGPU.cuh:579-583) on the same inputs,
 * and hmd_nlist_tiles_to_rows reproduces the reference's nlist / n_neigh arrays from them.
 *
 * The cell grid is cut into blocks of 2x2x2 cells.  One workgroup stages the block's 4x4x4 region
 * of cells (periodic images already shifted, so no minimum image per pair) into LDS; candidate k of
 * the region is numbered by (region cell, slot in the cell list).  A particle's row holds 16-bit
 * region-local candidate numbers (tile slots), k < d_n_neigh[i], padded to a multiple of four entries
 * with the slot of a dummy far away.
 *
 * Row storage is neighbor-major within groups of 16 home particles: the members of a block's home
 * cells (cells in block order, members in slot order, ghosts included) are numbered h = 0, 1, ...;
 * group d_block_group[block] + h / 16 holds the rows of sixteen of them, cut into pieces of 16 entries
 * (32 bytes) and interleaved: entry k of member h sits at the 16-bit offset
 *     (((d_block_group[block] + h / 16) * row_pieces + k / 16) * 16 + h % 16) * 16 + k % 16,
 * so the four lanes that share a home particle in the force kernel read 32 consecutive bytes and a
 * wave (16 home particles) reads 512 consecutive bytes per trip, whatever the lengths of the rows.
 * hmd_nlist_tiles_storage_entries() gives the size of d_tile_nlist.
 * The numbering is fixed by d_cell_size / d_cell_idx, which must not change between the build and
 * its uses.  Requirements (hmd_nlist_tiles_supported): every periodic direction has >= 3 cells and a
 * region of 64 cells holds at most `capacity` particles.
 * ------------------------------------------------------------------------------------------ */
typedef struct
    {
    const unsigned int* d_cell_size; /* particles per cell (hoomd/CellList.h getCellSizeArray) */
    const unsigned int* d_cell_idx;  /* particle index at cli(slot, cell) (getIndexArray) */
    hmd_index3d_t ci;
    hmd_index2d_t cli;
    unsigned int capacity;           /* candidates per region the LDS tile may hold */
    /* Required by the pair-force entry points (the kernel stages a region with one round of coalesced
       loads from them), optional outputs of the build:
       d_cpos[cli(slot, cell)]      position (w = type word) of the member in that slot, refreshed by
                                    hmd_tile_refresh before every force launch;
       d_slot_meta[cli(slot, cell)] x = particle index, y = n_neigh, z and w unused;
                                    written by hmd_compute_nlist_tiles. */
    const hmd_scalar4* d_cpos;
    const hmd_uint4* d_slot_meta;
    /* Row storage (see above): first group of every block (number of blocks + 1 entries, written by
       hmd_compute_nlist_tiles) and the number of 16-entry pieces every row has room for. */
    const unsigned int* d_block_group;
    unsigned int row_pieces;
    /* 1 when the rows were laid out by type (two particle types: the neighbors of type 0, the dummy candidate up to a
       multiple of four entries, then the neighbors of type 1), as hmd_nlist_tiles_rows_by_type says of the build that
       wrote them; the pair-force entry points then choose parameters per chunk of four entries.  0 is always valid. */
    unsigned int rows_by_type;
    } hmd_tile_geom_t;

/* Number of 16-bit entries d_tile_nlist must hold for a cell grid whose cells hold n_members particles in all
   (owned + ghost) when rows have room for row_pieces * 16 entries; includes the slack the force kernel's
   prefetch may touch. */
size_t hmd_nlist_tiles_storage_entries(const hmd_index3d_t* ci, unsigned int n_members, unsigned int row_pieces);

/* d_slot_of[particle index] = cli(slot, cell) for every member of the cell list (owned and ghost). */
int hmd_tile_slots(const hmd_tile_geom_t* geom, unsigned int* d_slot_of, void* stream);
/* d_cpos[d_slot_of[i]] = d_pos[i] for i < n (n = owned + ghost particles). */
int hmd_tile_refresh(const hmd_scalar4* d_pos, const unsigned int* d_slot_of, hmd_scalar4* d_cpos, unsigned int n, void* stream);

/* Returns HMD_OK and the LDS capacity to use when the tile path can serve this cell grid, else
   HMD_ERR_UNSUPPORTED (use the row-based entry points).  cell_Nmax: slots per cell (cli.w).  With more than one
   particle type the rows carry the candidate's type beside its slot: at most 8 types, and the capacity is capped at
   2751 candidates per 64-cell region (a build that meets a fuller region raises the overflow word). */
int hmd_nlist_tiles_supported(const hmd_index3d_t* ci, const hmd_box_t* box, unsigned int cell_Nmax, unsigned int ntypes,
                              unsigned int* capacity_out);

typedef struct
    {
    hmd_tile_geom_t geom;
    const hmd_scalar4* d_cell_xyzf;       /* cell list contents as for hmd_compute_nlist_binned (w = particle index) */
    const hmd_uint2* d_cell_type_body;    /* may be NULL with a single type */
    const unsigned int* d_cell_adj;       /* 27-cell stencil */
    hmd_index2d_t cadji;
    unsigned short* d_tile_nlist;
    hmd_uint4* d_slot_meta;          /* optional output, see hmd_tile_geom_t */
    unsigned int* d_slot_of;         /* optional output with d_slot_meta and geom.d_cpos: d_slot_of[particle] = its slot
                                        (what hmd_tile_slots computes) and d_cpos[slot] = its position (hmd_tile_refresh) */
    unsigned int* d_n_neigh;
    hmd_scalar4* d_last_updated_pos;
    unsigned int* d_conditions;      /* per type: largest row length seen (atomicMax), as the row-based build */
    const unsigned int* d_Nmax;
    const size_t* d_head_list;
    const hmd_scalar4* d_pos;
    unsigned int N;                  /* owned particles; cell members with index >= N are ghosts */
    hmd_box_t box;
    const double* d_r_cut;           /* ntypes x ntypes */
    double r_buff;
    unsigned int ntypes;
    unsigned int* d_tile_overflow;   /* two words, cleared by the call: [0] set to 1 when a region exceeds geom.capacity or the rows
                                        need more room than tile_nlist_entries (rows are then invalid); [1] raised
                                        (atomicMax) to the largest region population, the capacity later calls on
                                        these rows need */
    size_t tile_nlist_entries;       /* size of d_tile_nlist in 16-bit entries (hmd_nlist_tiles_storage_entries) */
    /* Optional exclusions by particle index (the arrays hmd_update_exclusion_list fills): excluded partners are
       left out of the rows while they are built, so rows and d_n_neigh come out as the reference's filter pass
       (hoomd/md/NeighborListGPU.cu:158-223) leaves them.  d_n_ex_idx NULL: no exclusions. */
    const unsigned int* d_n_ex_idx;
    const unsigned int* d_ex_list_idx; /* entry o of particle i at exli(i, o) */
    hmd_index2d_t exli;
    } hmd_nlist_tiles_args_t;
int hmd_compute_nlist_tiles(const hmd_nlist_tiles_args_t* args, void* stream);

/* Expand tile rows into particle indices: d_nlist[d_head_list[i] + k] for k < d_n_neigh[i]
   (the reference's layout, hoomd/md/NeighborList.h:662-739). */
typedef struct
    {
    hmd_tile_geom_t geom;
    const unsigned short* d_tile_nlist;
    const unsigned int* d_n_neigh;
    const size_t* d_head_list;
    unsigned int* d_nlist;
    unsigned int N;
    hmd_box_t box;
    unsigned int ntypes;             /* as given to hmd_compute_nlist_tiles: rows built for more than one type carry the
                                        candidate's type in the top three bits of an entry, the slot in the low 13 */
    } hmd_tiles_to_rows_args_t;
int hmd_nlist_tiles_to_rows(const hmd_tiles_to_rows_args_t* args, void* stream);
/* Whether hmd_compute_nlist_tiles lays the rows of this build out by type (see hmd_tile_geom_t::rows_by_type): two
 * types, no body filter, orthorhombic box, at least three cells along every periodic direction.  No counterpart in the
 * reference, whose rows are in cell-walk order (hoomd/md/NeighborListGPUBinned.cu:168-287); the order within a row is
 * free there too. */
int hmd_nlist_tiles_rows_by_type(const hmd_nlist_tiles_args_t* args);

/* Pair forces from tile rows; args->d_nlist, d_head_list and size_neigh_list are ignored (geom locates the rows),
   threads_per_particle must be 0 or 4 (four lanes share a home particle), everything else as
   hmd_compute_pair_forces_*. */
int hmd_compute_pair_forces_lj_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom, const unsigned short* d_tile_nlist,
                                     const hmd_lj_params_t* d_params, void* stream);
int hmd_compute_pair_forces_example_tiles(const hmd_pair_args_t* args, const hmd_tile_geom_t* geom,
                                          const unsigned short* d_tile_nlist, const hmd_example_params_t* d_params, void* stream);

/* hmd_compute_pair_forces_lj_tiles with the pass of hmd_nve_step_two_one_check riding on it: the lane that finishes
   a particle's force (gpu_compute_pair_forces, hoomd/md/PotentialPairGPU.cuh:140-404) goes on with the second half of
   the time step (hoomd/md/TwoStepConstantVolumeGPU.cu:215-260), the first half of the next one (:102-155), the distance
   check of the new position (hoomd/md/NeighborListGPU.cu:42-84) and the slot-ordered copy, so a time step of an NVE
   run with one pair force is ONE kernel.  Forces (args->d_force) are stored as always; positions, velocities, images,
   the check's result and step->d_cpos come out bit-identical to hmd_compute_pair_forces_lj_tiles followed by
   hmd_nve_step_two_one_check with two.d_net_force = args->d_force.  Requirements: no virial, the whole grid of blocks
   (block_window_mode 0), the group is every particle (d_group_members NULL, group_size == args->N), no displacement
   limit, one.d_vel == two.d_vel, and step->d_cpos / d_slot_of set with d_cpos a buffer OTHER than geom->d_cpos (the
   kernel stages from geom->d_cpos while it writes the new positions; the caller swaps the two buffers afterwards).
   two.d_net_force and one.d_accel are not read; two.d_accel must be NULL (the accelerations are not stored).  The
   pass rides only in the case of a plain NVE run: orthorhombic box periodic in all directions and unchanged since the
   neighbor list was built (check.box == one.box, lambda == 1), rescale factors 1, one time step length;
   hmd_pair_forces_tiles_step_supported returns HMD_OK when it can ride and HMD_ERR_UNSUPPORTED (reason in
   hmd_last_error()) when the caller has to run the two entry points one after the other.  A launch that
   args->d_skip_flag switches off leaves everything untouched. */
int hmd_pair_forces_tiles_step_supported(const hmd_pair_args_t* args, const hmd_fused_step_args_t* step);
int hmd_compute_pair_forces_lj_tiles_step(const hmd_pair_args_t* args,
                                          const hmd_tile_geom_t* geom,
                                          const unsigned short* d_tile_nlist,
                                          const hmd_lj_params_t* d_params,
                                          const hmd_fused_step_args_t* step,
                                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOOMD_AMD_H */
