// pair_force.cuh -- pair-force kernel template for gfx950.
//
// Contract: hoomd/md/PotentialPairGPU.cuh:140-404 (kernel semantics) and :579-656 (driver);
// per-pair arithmetic order as hoomd/md/PotentialPair.h:660-752.  Results agree with the CPU path
// to rounding (different summation order, fused multiply-adds allowed), well inside 1e-6 relative.
//
// Design: TPP lanes share one particle.  Per loop trip each lane takes FOUR consecutive entries of
// the particle's padded neighbor row with one 16-byte load (the TPP lanes of a particle read
// TPP*16 contiguous bytes), issues the four pos[j] gathers back to back so that eight 16-byte loads
// are in flight per lane, prefetches the next trip's index slab, and only then evaluates the four
// pairs.  The partial sums are folded with TPP-wide butterfly shuffles.  Type-pair
// tables (params, r_cut^2, r_on^2) sit in LDS; with a single particle type they are wave-uniform
// scalars and the type lookup disappears.  Workgroup ids are remapped so that each XCD walks a
// contiguous range of particles: with the arrays in cell order its gathers stay inside one L2.
#ifndef HMD_PAIR_FORCE_CUH
#define HMD_PAIR_FORCE_CUH

#include "common.cuh"
#include <type_traits>
#include <utility>

namespace hmd
    {
constexpr int PF_BLOCK = 256;

// The kernels are templates over an evaluator class of the reference's shape (hoomd/md/EvaluatorPairLJ.h:35-81,
// example_plugins/pair_plugin/EvaluatorPairExample.h:33-167):
//   struct param_type            plain data, one per type pair, copied to LDS by the kernel
//   Evaluator(rsq, rcutsq, const param_type&)
//   static bool needsCharge();   void setCharge(qi, qj);
//   bool evalForceAndEnergy(Scalar& force_divr, Scalar& pair_eng, bool energy_shift);
// Optionally an evaluator offers evalUnshiftedBranchFree(force_divr, pair_eng) const: force and energy of the unshifted
// potential that vanish by themselves outside the cutoff; the kernels use it when it exists.
template<class Ev, class = void> struct has_branch_free
    {
    static constexpr bool value = false;
    };
template<class Ev> struct has_branch_free<Ev, decltype(std::declval<const Ev&>().evalUnshiftedBranchFree(std::declval<double&>(), std::declval<double&>()))>
    {
    static constexpr bool value = true;
    };

// Optionally, too, the static evalUnshiftedBranchFree4(rsq[4], rcutsq[4], params[4], force_divr[4], pair_eng[4]): the same for four
// pairs at once, which lets an evaluator share work between them (one reciprocal for four pairs in the LJ evaluator).
template<class Ev, class = void> struct has_branch_free4
    {
    static constexpr bool value = false;
    };
template<class Ev>
struct has_branch_free4<Ev, decltype(Ev::evalUnshiftedBranchFree4(std::declval<const double (&)[4]>(), std::declval<const double (&)[4]>(),
                                                                 std::declval<const typename Ev::param_type* const (&)[4]>(), std::declval<double (&)[4]>(),
                                                                 std::declval<double (&)[4]>()))>
    {
    static constexpr bool value = true;
    };

__device__ __forceinline__ unsigned int pf_xcd_remap(unsigned int b, unsigned int nb)
    {
    const unsigned int q = nb / 8, r = nb % 8, x = b % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
    }

// value of `v` in another lane through the data-parallel-primitive path of the VALU (no LDS traffic,
// unlike __shfl*, which goes through ds_bpermute)
template<int CTRL> __device__ __forceinline__ double pf_dpp(double v)
    {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
    }

// Sum over the TPP lanes of a particle; the result is complete in the group's first lane (all lanes for
// TPP > 16).  quad_perm [1,0,3,2] = 0xb1, [2,3,0,1] = 0x4e, row_shl:4 = 0x104, row_shl:8 = 0x108.
template<int TPP> __device__ __forceinline__ double pf_reduce(double v)
    {
    if (TPP > 16)
        {
#pragma unroll
        for (int m = TPP / 2; m > 0; m >>= 1)
            v += __shfl_xor(v, m, 64);
        return v;
        }
    if (TPP >= 2)
        v += pf_dpp<0xb1>(v);
    if (TPP >= 4)
        v += pf_dpp<0x4e>(v);
    if (TPP >= 8)
        v += pf_dpp<0x104>(v);
    if (TPP >= 16)
        v += pf_dpp<0x108>(v);
    return v;
    }

// Box terms of the minimum-image step, hoisted out of the pair loop.  A non-periodic direction gets
// Linv = 0, so its image count is rint(0) = 0 and no select is needed per pair.
struct pf_box_t
    {
    double Lx, Ly, Lz, Lix, Liy, Liz;
    double Lz_yz, Lz_xz, Ly_xy;
    };

__device__ __forceinline__ pf_box_t pf_make_box(const hmd_box_t& b)
    {
    pf_box_t o;
    o.Lx = b.L[0];
    o.Ly = b.L[1];
    o.Lz = b.L[2];
    o.Lix = b.periodic[0] ? b.Linv[0] : 0.0;
    o.Liy = b.periodic[1] ? b.Linv[1] : 0.0;
    o.Liz = b.periodic[2] ? b.Linv[2] : 0.0;
    o.Lz_yz = b.L[2] * b.yz;
    o.Lz_xz = b.L[2] * b.xz;
    o.Ly_xy = b.L[1] * b.xy;
    return o;
    }

// minimum image for the force loop (hoomd/BoxDim.h:306-325); fused multiply-adds allowed
template<bool ORTHO> __device__ __forceinline__ void pf_min_image(const pf_box_t& b, double& x, double& y, double& z)
    {
    const double iz = rint(z * b.Liz);
    z -= b.Lz * iz;
    if (!ORTHO)
        {
        y -= b.Lz_yz * iz;
        x -= b.Lz_xz * iz;
        }
    const double iy = rint(y * b.Liy);
    y -= b.Ly * iy;
    if (!ORTHO)
        x -= b.Ly_xy * iy;
    x -= b.Lx * rint(x * b.Lix);
    }

// running sums of one lane
struct pf_accum
    {
    double fx = 0, fy = 0, fz = 0, pe = 0;
    double vxx = 0, vxy = 0, vxz = 0, vyy = 0, vyz = 0, vzz = 0;
    };

// One pair: evaluator, energy shift / XPLOR switching and accumulation
// (hoomd/md/PotentialPair.h:660-752, hoomd/md/PotentialPairGPU.cuh:300-372).
template<class Ev, unsigned int SHIFT, bool VIRIAL>
__device__ __forceinline__ void pf_pair(pf_accum& s, double dx, double dy, double dz, double rcutsq, double ronsq,
                                        const typename Ev::param_type& prm, double qi, double qj)
    {
    const double rsq = dx * dx + dy * dy + dz * dz;
    bool energy_shift = false;
    if (SHIFT == 1)
        energy_shift = true;
    else if (SHIFT == 2)
        energy_shift = ronsq > rcutsq;

    double fdr = 0.0, eng = 0.0;
    Ev ev(rsq, rcutsq, prm);
    if (Ev::needsCharge())
        ev.setCharge(qi, qj);
    bool evaluated = true;
    if constexpr (SHIFT == 0 && has_branch_free<Ev>::value)
        ev.evalUnshiftedBranchFree(fdr, eng);
    else
        {
        evaluated = ev.evalForceAndEnergy(fdr, eng, energy_shift);
        if (!evaluated)
            {
            fdr = 0.0;
            eng = 0.0;
            }
        }
    if (SHIFT == 2)
        {
        if (evaluated && rsq >= ronsq && rsq < rcutsq)
            {
            // XPLOR switching, hoomd/md/PotentialPair.h:711-737
            const double d3 = (rcutsq - ronsq) * (rcutsq - ronsq) * (rcutsq - ronsq);
            const double denom_inv = 1.0 / d3;
            const double d = rsq - rcutsq;
            const double sw = d * d * (rcutsq + 2.0 * rsq - 3.0 * ronsq) * denom_inv;
            const double dsw = 12.0 * (rsq - ronsq) * d * denom_inv;
            fdr = sw * fdr - dsw * eng;
            eng = eng * sw;
            }
        }
    s.fx += dx * fdr;
    s.fy += dy * fdr;
    s.fz += dz * fdr;
    s.pe += eng;
    if (VIRIAL)
        {
        const double f2 = 0.5 * fdr;
        s.vxx += dx * dx * f2;
        s.vxy += dx * dy * f2;
        s.vxz += dx * dz * f2;
        s.vyy += dy * dy * f2;
        s.vyz += dy * dz * f2;
        s.vzz += dz * dz * f2;
        }
    }

// Four pairs of one home particle through the evaluator's four-pair form (unshifted potential, no charges)
template<class Ev, bool VIRIAL>
__device__ __forceinline__ void pf_pair4(pf_accum& s, const double (&dx)[4], const double (&dy)[4], const double (&dz)[4], const double (&rcutsq)[4],
                                         const typename Ev::param_type* const (&prm)[4])
    {
    double rsq[4], fdr[4], eng[4];
#pragma unroll
    for (int m = 0; m < 4; m++)
        rsq[m] = dx[m] * dx[m] + dy[m] * dy[m] + dz[m] * dz[m];
    Ev::evalUnshiftedBranchFree4(rsq, rcutsq, prm, fdr, eng);
#pragma unroll
    for (int m = 0; m < 4; m++)
        {
        s.fx += dx[m] * fdr[m];
        s.fy += dy[m] * fdr[m];
        s.fz += dz[m] * fdr[m];
        s.pe += eng[m];
        if (VIRIAL)
            {
            const double f2 = 0.5 * fdr[m];
            s.vxx += dx[m] * dx[m] * f2;
            s.vxy += dx[m] * dy[m] * f2;
            s.vxz += dx[m] * dz[m] * f2;
            s.vyy += dy[m] * dy[m] * f2;
            s.vyz += dy[m] * dz[m] * f2;
            s.vzz += dz[m] * dz[m] * f2;
            }
        }
    }

// fold the TPP lanes of a particle and store force (w = half the pair energy) and virial
// (returns the force, complete in the writer's lane)
template<bool VIRIAL, int TPP>
__device__ __forceinline__ double4 pf_finish(pf_accum& s, const hmd_pair_args_t& a, unsigned int idx, bool writer)
    {
    const double fx = pf_reduce<TPP>(s.fx), fy = pf_reduce<TPP>(s.fy), fz = pf_reduce<TPP>(s.fz);
    const double pe = 0.5 * pf_reduce<TPP>(s.pe);
    if (writer)
        hmd_st4(a.d_force + idx, fx, fy, fz, pe);
    if (VIRIAL)
        {
        const double vxx = pf_reduce<TPP>(s.vxx), vxy = pf_reduce<TPP>(s.vxy), vxz = pf_reduce<TPP>(s.vxz);
        const double vyy = pf_reduce<TPP>(s.vyy), vyz = pf_reduce<TPP>(s.vyz), vzz = pf_reduce<TPP>(s.vzz);
        if (writer)
            {
            a.d_virial[0 * a.virial_pitch + idx] = vxx;
            a.d_virial[1 * a.virial_pitch + idx] = vxy;
            a.d_virial[2 * a.virial_pitch + idx] = vxz;
            a.d_virial[3 * a.virial_pitch + idx] = vyy;
            a.d_virial[4 * a.virial_pitch + idx] = vyz;
            a.d_virial[5 * a.virial_pitch + idx] = vzz;
            }
        }
    return make_double4(fx, fy, fz, pe);
    }

constexpr int PF_VEC = 4; // neighbors handled per lane per loop trip (one 16-byte index load)

template<class Ev, unsigned int SHIFT, bool VIRIAL, int TPP, bool SINGLE_TYPE, bool ORTHO>
__global__ void __launch_bounds__(PF_BLOCK) pair_force_kernel(hmd_pair_args_t a, const typename Ev::param_type* d_params)
    {
    typedef typename Ev::param_type param_t;
    if (a.d_skip_flag)
        {
        const unsigned int word = *a.d_skip_flag;
        if (a.d_published && blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(a.d_published, ((unsigned long long)a.publish_stamp << 32) | (unsigned long long)word, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        if (word == a.skip_value)
            return;
        }
    extern __shared__ __align__(16) unsigned char s_raw[];
    const unsigned int np = a.ntypes * a.ntypes;
    param_t* s_params = reinterpret_cast<param_t*>(s_raw);
    double* s_rcutsq = reinterpret_cast<double*>(s_raw + (sizeof(param_t) * np + 15) / 16 * 16);
    double* s_ronsq = s_rcutsq + np;

    param_t p0;
    double rcutsq0 = 0.0, ronsq0 = 0.0;
    if (SINGLE_TYPE)
        {
        p0 = d_params[0];
        rcutsq0 = a.d_rcutsq[0];
        if (SHIFT == 2)
            ronsq0 = a.d_ronsq[0];
        }
    else
        {
        for (unsigned int t = threadIdx.x; t < np; t += PF_BLOCK)
            {
            s_params[t] = d_params[t];
            s_rcutsq[t] = a.d_rcutsq[t];
            s_ronsq[t] = (SHIFT == 2) ? a.d_ronsq[t] : 0.0;
            }
        __syncthreads();
        }
    const pf_box_t box = pf_make_box(a.box);

    constexpr unsigned int PPB = PF_BLOCK / TPP; // particles per workgroup
    const unsigned int vb = pf_xcd_remap(blockIdx.x, gridDim.x);
    const unsigned int idx = vb * PPB + threadIdx.x / TPP;
    const unsigned int sub = threadIdx.x % TPP;
    const bool active = idx < a.N;

    pf_accum acc;

    if (active)
        {
        const unsigned int nn = a.d_n_neigh[idx];
        const double4 pi = hmd_ld4(a.d_pos + idx);
        const unsigned int ti = SINGLE_TYPE ? 0u : hmd_double_as_uint(pi.w);
        double qi = 0.0;
        if (Ev::needsCharge())
            qi = a.d_charge[idx];
        const size_t head = a.d_head_list[idx];
        const unsigned int* row = a.d_nlist + head;
        // rows that start on a 16-byte boundary and are padded to a multiple of four entries (what the
        // neighbor-list build produces) are read four indices at a time
        const bool vec_ok = ((head & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.d_nlist) & 15) == 0) && (a.size_neigh_list >= head + ((nn + 3) & ~3u));

        // Lane `sub` of the particle's TPP lanes takes entries kb + m*TPP + sub, m = 0..3, of each
        // 4*TPP-entry slab: neighboring lanes gather neighboring list entries, which in a cell-sorted
        // system are mostly neighboring particles -- their 32-byte positions share cache lines.
        auto load_slab = [&](unsigned int kb, unsigned int (&v)[PF_VEC])
        {
#pragma unroll
            for (int m = 0; m < PF_VEC; m++)
                {
                const unsigned int e = kb + m * TPP + sub;
                v[m] = (e < nn) ? row[e] : idx;
                }
        };
        (void)vec_ok;
        unsigned int cur[PF_VEC], nxt[PF_VEC];
        load_slab(0, cur);
        for (unsigned int kb = 0; kb < nn; kb += TPP * PF_VEC)
            {
            // index slab of the next trip: in flight while this trip's positions are gathered
            load_slab(kb + TPP * PF_VEC, nxt);
            double4 pj[PF_VEC];
#pragma unroll
            for (int m = 0; m < PF_VEC; m++)
                pj[m] = hmd_ld4(a.d_pos + cur[m]);
#pragma unroll
            for (int m = 0; m < PF_VEC; m++)
                {
                double dx = pi.x - pj[m].x, dy = pi.y - pj[m].y, dz = pi.z - pj[m].z;
                pf_min_image<ORTHO>(box, dx, dy, dz);
                double rcutsq, ronsq = 0.0;
                const param_t* prm;
                if (SINGLE_TYPE)
                    {
                    rcutsq = rcutsq0;
                    ronsq = ronsq0;
                    prm = &p0;
                    }
                else
                    {
                    const unsigned int tp = hmd_double_as_uint(pj[m].w) * a.ntypes + ti;
                    rcutsq = s_rcutsq[tp];
                    if (SHIFT == 2)
                        ronsq = s_ronsq[tp];
                    prm = s_params + tp;
                    }
                // the padding entries (j == idx) sit at rsq = 0; disable them through the cutoff
                if (kb + m * TPP + sub >= nn)
                    rcutsq = -1.0;
                double qj = 0.0;
                if (Ev::needsCharge())
                    qj = a.d_charge[cur[m]];
                pf_pair<Ev, SHIFT, VIRIAL>(acc, dx, dy, dz, rcutsq, ronsq, *prm, qi, qj);
                }
#pragma unroll
            for (int m = 0; m < PF_VEC; m++)
                cur[m] = nxt[m];
            }
        }
    pf_finish<VIRIAL, TPP>(acc, a, idx, active && sub == 0);
    }

template<class Ev, unsigned int SHIFT, bool VIRIAL, int TPP>
int pf_launch3(const hmd_pair_args_t& a, const typename Ev::param_type* d_params, hipStream_t s)
    {
    const unsigned int np = a.ntypes * a.ntypes;
    const bool single = a.ntypes == 1;
    const size_t lds = single ? 0 : (sizeof(typename Ev::param_type) * np + 15) / 16 * 16 + 2 * sizeof(double) * np;
    const dim3 grid(hmd_div_up(a.N, PF_BLOCK / TPP));
    const bool ortho = hmd_box_is_ortho(a.box);
    if (single && ortho)
        hipLaunchKernelGGL((pair_force_kernel<Ev, SHIFT, VIRIAL, TPP, true, true>), grid, dim3(PF_BLOCK), lds, s, a, d_params);
    else if (single)
        hipLaunchKernelGGL((pair_force_kernel<Ev, SHIFT, VIRIAL, TPP, true, false>), grid, dim3(PF_BLOCK), lds, s, a, d_params);
    else if (ortho)
        hipLaunchKernelGGL((pair_force_kernel<Ev, SHIFT, VIRIAL, TPP, false, true>), grid, dim3(PF_BLOCK), lds, s, a, d_params);
    else
        hipLaunchKernelGGL((pair_force_kernel<Ev, SHIFT, VIRIAL, TPP, false, false>), grid, dim3(PF_BLOCK), lds, s, a, d_params);
    return hmd_launch_status();
    }

template<class Ev, unsigned int SHIFT, bool VIRIAL>
int pf_launch2(const hmd_pair_args_t& a, const typename Ev::param_type* d_params, unsigned int tpp, hipStream_t s)
    {
    switch (tpp)
        {
    case 1:
        return pf_launch3<Ev, SHIFT, VIRIAL, 1>(a, d_params, s);
    case 2:
        return pf_launch3<Ev, SHIFT, VIRIAL, 2>(a, d_params, s);
    case 4:
        return pf_launch3<Ev, SHIFT, VIRIAL, 4>(a, d_params, s);
    case 8:
        return pf_launch3<Ev, SHIFT, VIRIAL, 8>(a, d_params, s);
    case 16:
        return pf_launch3<Ev, SHIFT, VIRIAL, 16>(a, d_params, s);
    case 32:
        return pf_launch3<Ev, SHIFT, VIRIAL, 32>(a, d_params, s);
    default:
        return pf_launch3<Ev, SHIFT, VIRIAL, 64>(a, d_params, s);
        }
    }

// needsCharge() of an evaluator as the host sees it.  The contract declares it a device function (an evaluator written
// for the reference has it so), which host code cannot call: a one-thread kernel asks once per evaluator and process.
// If the question cannot be asked the answer is "yes" (the charge array is then required, which is always safe).
static __device__ int pf_query_word;
template<class Ev> __global__ void pf_needs_charge_kernel()
    {
    pf_query_word = Ev::needsCharge() ? 1 : 0;
    }
template<class Ev> bool pf_needs_charge(hipStream_t s)
    {
    static const bool answer = [s]()
    {
        int h = 1;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
            return true;
        hipLaunchKernelGGL(pf_needs_charge_kernel<Ev>, dim3(1), dim3(1), 0, s);
        if (hipMemcpyFromSymbolAsync(&h, HIP_SYMBOL(pf_query_word), sizeof(int), 0, hipMemcpyDeviceToHost, s) != hipSuccess
            || hipStreamSynchronize(s) != hipSuccess)
            return true;
        return h != 0;
    }();
    return answer;
    }

// Driver shared by every evaluator's extern "C" entry point.
template<class Ev> int launch_pair_forces(const hmd_pair_args_t* args, const typename Ev::param_type* d_params, void* stream, const char* who)
    {
    HMD_REQUIRE(args && d_params, HMD_ERR_INVALID_ARGUMENT, "pair forces: args and d_params are required");
    const hmd_pair_args_t& a = *args;
    HMD_REQUIRE(a.d_force && a.d_pos && a.d_n_neigh && a.d_nlist && a.d_head_list && a.d_rcutsq, HMD_ERR_INVALID_ARGUMENT,
                "pair forces: force, pos, n_neigh, nlist, head_list and rcutsq are required");
    HMD_REQUIRE(a.shift_mode <= 2, HMD_ERR_INVALID_ARGUMENT, "pair forces: shift_mode must be 0 (none), 1 (shift) or 2 (xplor)");
    HMD_REQUIRE(a.shift_mode != 2 || a.d_ronsq, HMD_ERR_INVALID_ARGUMENT, "pair forces: xplor needs d_ronsq");
    HMD_REQUIRE(!a.compute_virial || a.d_virial, HMD_ERR_INVALID_ARGUMENT, "pair forces: compute_virial needs d_virial");
    HMD_REQUIRE(a.d_charge || !pf_needs_charge<Ev>((hipStream_t)stream), HMD_ERR_INVALID_ARGUMENT, "pair forces: this evaluator needs d_charge");
    HMD_REQUIRE(a.ntypes >= 1, HMD_ERR_INVALID_ARGUMENT, "pair forces: ntypes must be >= 1");
    unsigned int tpp = a.threads_per_particle ? a.threads_per_particle : 8;
    HMD_REQUIRE(tpp <= 64 && (tpp & (tpp - 1)) == 0, HMD_ERR_INVALID_ARGUMENT,
                "pair forces: threads_per_particle must be a power of two in [1, 64]");
    const size_t lds = (sizeof(typename Ev::param_type) + 16) * (size_t)a.ntypes * a.ntypes + 16;
    HMD_REQUIRE(lds <= 64 * 1024, HMD_ERR_TOO_LARGE, "pair forces: type-pair tables exceed 64 KiB of LDS");
    (void)who;
    if (a.N == 0)
        return HMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool vir = a.compute_virial != 0;
    switch (a.shift_mode)
        {
    case 0:
        return vir ? pf_launch2<Ev, 0, true>(a, d_params, tpp, s) : pf_launch2<Ev, 0, false>(a, d_params, tpp, s);
    case 1:
        return vir ? pf_launch2<Ev, 1, true>(a, d_params, tpp, s) : pf_launch2<Ev, 1, false>(a, d_params, tpp, s);
    default:
        return vir ? pf_launch2<Ev, 2, true>(a, d_params, tpp, s) : pf_launch2<Ev, 2, false>(a, d_params, tpp, s);
        }
    }
    } // namespace hmd

#endif
