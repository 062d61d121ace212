// bond_force.hip -- FENE + WCA bond forces and the per-particle bond table.
// Replaces gpu_compute_bond_forces<EvaluatorBondFENE, 2> (hoomd/md/PotentialBondGPU.cuh:97-330 with
// hoomd/md/EvaluatorBondFENE.h:116-156) and gpu_update_group_table<2> (hoomd/BondedGroupData.cu:125-220).
#include "common.cuh"

namespace
    {
// hoomd/md/EvaluatorBondFENE.h:90-94, :116-156
struct EvaluatorBondFENE
    {
    __device__ __forceinline__ EvaluatorBondFENE(double rsq_, const hmd_fene_params_t& p)
        : rsq(rsq_), K(p.k), r_0(p.r_0), lj1(p.epsilon_x_4 * p.sigma_6 * p.sigma_6), lj2(p.epsilon_x_4 * p.sigma_6), delta(p.delta)
        {
        }
    __device__ __forceinline__ bool evalForceAndEnergy(double& force_divr, double& bond_eng)
        {
        const double r = sqrt(rsq);
        const double rtemp = r - delta;
        const double rmdoverr = rtemp / r;
        rsq = rtemp * rtemp;
        const double r2inv = 1.0 / rsq;
        const double r6inv = r2inv * r2inv * r2inv;
        double wca_divr = 0.0, pair_eng = 0.0;
        const double sigma6inv = lj2 / lj1;
        const double epsilon = lj2 * lj2 / 4.0 / lj1;
        if (lj1 != 0 && r6inv > sigma6inv / 2.0)
            {
            wca_divr = r2inv * r6inv * (12.0 * lj1 * r6inv - 6.0 * lj2);
            pair_eng = r6inv * (lj1 * r6inv - lj2) + epsilon;
            }
        force_divr = wca_divr * rmdoverr;
        bond_eng = pair_eng;
        if (K != 0.0)
            {
            if (rsq >= r_0 * r_0)
                return false;
            force_divr += -K / (1.0 - rsq / (r_0 * r_0)) * rmdoverr;
            bond_eng += -0.5 * K * (r_0 * r_0) * log(1.0 - rsq / (r_0 * r_0));
            }
        return true;
        }
    double rsq, K, r_0, lj1, lj2, delta;
    };

template<bool VIRIAL, bool ORTHO> __global__ void __launch_bounds__(256) bond_force_fene_kernel(hmd_bond_args_t a, const hmd_fene_params_t* d_params)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const unsigned int n_bonds = a.d_n_bonds[idx];
    const double4 pi = hmd_ld4(a.d_pos + idx);
    double fx = 0, fy = 0, fz = 0, pe = 0;
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned int k = 0; k < n_bonds; k++)
        {
        const hmd_uint2 b = a.d_blist[hmd_idx2(a.blist_idx, idx, k)];
        const double4 pj = hmd_ld4(a.d_pos + b.x);
        double dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        hmd_min_image<ORTHO>(a.box, dx, dy, dz);
        const double rsq = dx * dx + dy * dy + dz * dz;
        double force_divr = 0.0, bond_eng = 0.0;
        EvaluatorBondFENE eval(rsq, d_params[b.y]);
        if (!eval.evalForceAndEnergy(force_divr, bond_eng))
            {
            *a.d_flags = 1u;
            return;
            }
        if (VIRIAL)
            {
            const double f2 = force_divr / 2.0;
            v[0] += dx * dx * f2;
            v[1] += dx * dy * f2;
            v[2] += dx * dz * f2;
            v[3] += dy * dy * f2;
            v[4] += dy * dz * f2;
            v[5] += dz * dz * f2;
            }
        fx += dx * force_divr;
        fy += dy * force_divr;
        fz += dz * force_divr;
        pe += bond_eng * 0.5;
        }
    hmd_st4(a.d_force + idx, fx, fy, fz, pe);
    if (VIRIAL)
        for (int c = 0; c < 6; c++)
            a.d_virial[c * a.virial_pitch + idx] = v[c];
    }

__global__ void __launch_bounds__(256) bond_table_kernel(hmd_bond_table_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N)
        return;
    const unsigned int tag = a.d_tag[idx];
    const unsigned int n = min(a.d_tag_n_bonds[tag], a.blist_idx.h);
    a.d_n_bonds[idx] = n;
    for (unsigned int k = 0; k < n; k++)
        {
        hmd_uint2 e = a.d_tag_blist[hmd_idx2(a.tag_blist_idx, tag, k)];
        const unsigned int partner = a.d_rtag[e.x];
        if (partner >= a.n_total)
            {
            *a.d_flags = 1u;
            a.d_n_bonds[idx] = k; // nothing after the incomplete bond is trusted
            return;
            }
        e.x = partner;
        a.d_blist[hmd_idx2(a.blist_idx, idx, k)] = e;
        }
    }
    } // namespace

extern "C" int hmd_update_bond_table(const hmd_bond_table_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_update_bond_table: args is NULL");
    const hmd_bond_table_args_t& a = *args;
    HMD_REQUIRE(a.d_tag && a.d_rtag && a.d_tag_n_bonds && a.d_n_bonds && a.d_flags, HMD_ERR_INVALID_ARGUMENT,
                "hmd_update_bond_table: tag, rtag, tag_n_bonds, n_bonds and flags are required");
    HMD_REQUIRE(a.blist_idx.h == 0 || (a.d_tag_blist && a.d_blist), HMD_ERR_INVALID_ARGUMENT, "hmd_update_bond_table: the bond tables are required");
    HMD_REQUIRE(a.blist_idx.w >= a.N, HMD_ERR_INVALID_ARGUMENT, "hmd_update_bond_table: blist_idx.w must be at least N");
    HMD_REQUIRE(a.tag_blist_idx.h >= a.blist_idx.h, HMD_ERR_INVALID_ARGUMENT, "hmd_update_bond_table: the index table holds fewer bonds per particle than the tag table");
    HMD_REQUIRE(a.n_total >= a.N, HMD_ERR_INVALID_ARGUMENT, "hmd_update_bond_table: n_total < N");
    if (a.N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(bond_table_kernel, dim3(hmd_div_up(a.N, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hmd_launch_status();
    }

extern "C" int hmd_compute_bond_forces_fene(const hmd_bond_args_t* args, const hmd_fene_params_t* d_params, void* stream)
    {
    HMD_REQUIRE(args && d_params, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_bond_forces_fene: args and d_params are required");
    const hmd_bond_args_t& a = *args;
    HMD_REQUIRE(a.d_force && a.d_pos && a.d_n_bonds && a.d_flags, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_bond_forces_fene: force, pos, n_bonds and flags are required");
    HMD_REQUIRE(a.blist_idx.h == 0 || a.d_blist, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_bond_forces_fene: d_blist is required");
    HMD_REQUIRE(a.blist_idx.w >= a.N, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_bond_forces_fene: blist_idx.w must be at least N");
    HMD_REQUIRE(!a.compute_virial || (a.d_virial && a.virial_pitch >= a.N), HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_bond_forces_fene: compute_virial needs d_virial with a pitch of at least N");
    HMD_REQUIRE(a.n_bond_types >= 1, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_bond_forces_fene: no bond types");
    if (a.N == 0)
        return HMD_OK;
    const bool ortho = a.box.xy == 0.0 && a.box.xz == 0.0 && a.box.yz == 0.0;
    const dim3 grid(hmd_div_up(a.N, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a.compute_virial)
        {
        if (ortho)
            hipLaunchKernelGGL((bond_force_fene_kernel<true, true>), grid, block, 0, s, a, d_params);
        else
            hipLaunchKernelGGL((bond_force_fene_kernel<true, false>), grid, block, 0, s, a, d_params);
        }
    else
        {
        if (ortho)
            hipLaunchKernelGGL((bond_force_fene_kernel<false, true>), grid, block, 0, s, a, d_params);
        else
            hipLaunchKernelGGL((bond_force_fene_kernel<false, false>), grid, block, 0, s, a, d_params);
        }
    return hmd_launch_status();
    }
