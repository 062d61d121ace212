// integrator.hip -- velocity-Verlet half steps, net-force sum, accelerations and thermo sums.
//
// Contracts: hoomd/md/TwoStepConstantVolumeGPU.cu:35-260 (drivers), arithmetic of the CPU class
// hoomd/md/TwoStepConstantVolume.cc:31-79,233-261 evaluated without fused multiply-adds, so one
// half step reproduces the CPU path bit for bit; hoomd/Integrator.cu:67-230 (net force);
// hoomd/Integrator.cc:106-127 (accelerations); hoomd/md/ComputeThermoGPU.cu (reductions).
// All of these stream each array once: 16-byte vector accesses, one lane per particle.
// Results of this unit are compared bit for bit with the CPU path: no product may be fused into a
// following sum (the CPU path rounds every product and sum separately).
#pragma clang fp contract(off)
#include "common.cuh"
#include "step_body.cuh"

namespace
    {
__global__ void __launch_bounds__(256) step_one_kernel(hmd_step_one_args_t a)
    {
    const unsigned int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.group_size)
        return;
    const unsigned int idx = a.d_group_members ? a.d_group_members[g] : g;

    const double4 p = hmd_ld4(a.d_pos + idx);
    const double4 v = hmd_ld4(a.d_vel + idx);
    const hmd_scalar3 acc = a.d_accel[idx];

    // v = v + (1/2 a) dt ; v *= rescale
    const double rs = hmd_rescale_factor(a);
    double vx = HMD_MUL(HMD_ADD(v.x, HMD_MUL(HMD_MUL(0.5, acc.x), a.deltaT)), rs);
    double vy = HMD_MUL(HMD_ADD(v.y, HMD_MUL(HMD_MUL(0.5, acc.y), a.deltaT)), rs);
    double vz = HMD_MUL(HMD_ADD(v.z, HMD_MUL(HMD_MUL(0.5, acc.z), a.deltaT)), rs);

    if (a.use_limit)
        {
        // hoomd/md/TwoStepConstantVolume.cc:45-53
        const double len = HMD_MUL(sqrt(HMD_ADD(HMD_ADD(HMD_MUL(vx, vx), HMD_MUL(vy, vy)), HMD_MUL(vz, vz))), a.deltaT);
        if (len > a.maximum_displacement)
            {
            vx = vx / len * a.maximum_displacement / a.deltaT;
            vy = vy / len * a.maximum_displacement / a.deltaT;
            vz = vz / len * a.maximum_displacement / a.deltaT;
            }
        }

    double x = HMD_ADD(p.x, HMD_MUL(a.deltaT, vx));
    double y = HMD_ADD(p.y, HMD_MUL(a.deltaT, vy));
    double z = HMD_ADD(p.z, HMD_MUL(a.deltaT, vz));

    // few particles cross the box boundary in a step: touch the image only for those
    hmd_int3 dimg = {0, 0, 0};
    hmd_wrap(a.box, x, y, z, dimg.x, dimg.y, dimg.z);

    hmd_st4(a.d_pos + idx, x, y, z, p.w);
    hmd_st4(a.d_vel + idx, vx, vy, vz, v.w);
    if (dimg.x | dimg.y | dimg.z)
        {
        hmd_int3 img = a.d_image[idx];
        img.x += dimg.x;
        img.y += dimg.y;
        img.z += dimg.z;
        a.d_image[idx] = img;
        }
    }

// the second half step of one particle; returns the new velocity (w = mass) and the particle's potential energy
__device__ __forceinline__ double4 step_two_particle(const hmd_step_two_args_t& a, unsigned int idx, double& pe)
    {
    double4 f = hmd_ld4(a.d_net_force + idx);
    if (a.d_net_force2)
        {
        // the sum of two forces, as the net-force pass computes it (0 + f1 is f1)
        const double4 f2 = hmd_ld4(a.d_net_force2 + idx);
        f.x = HMD_ADD(f.x, f2.x);
        f.y = HMD_ADD(f.y, f2.y);
        f.z = HMD_ADD(f.z, f2.z);
        }
    const double4 v = hmd_ld4(a.d_vel + idx);
    const double minv = 1.0 / v.w;
    hmd_scalar3 acc;
    acc.x = HMD_MUL(f.x, minv);
    acc.y = HMD_MUL(f.y, minv);
    acc.z = HMD_MUL(f.z, minv);
    const double hdt = HMD_MUL(0.5, a.deltaT);
    const double vx = HMD_ADD(HMD_MUL(v.x, a.rescale_factor), HMD_MUL(hdt, acc.x));
    const double vy = HMD_ADD(HMD_MUL(v.y, a.rescale_factor), HMD_MUL(hdt, acc.y));
    const double vz = HMD_ADD(HMD_MUL(v.z, a.rescale_factor), HMD_MUL(hdt, acc.z));
    hmd_st4(a.d_vel + idx, vx, vy, vz, v.w);
    a.d_accel[idx] = acc;
    pe = f.w;
    return make_double4(vx, vy, vz, v.w);
    }

__global__ void __launch_bounds__(256) step_two_kernel(hmd_step_two_args_t a)
    {
    const unsigned int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.group_size)
        return;
    double pe;
    step_two_particle(a, a.d_group_members ? a.d_group_members[g] : g, pe);
    }

// step two of t, step one of t+1 and the distance check in one pass (same arithmetic, same order, as the
// three kernels they replace: step_two_kernel, step_one_kernel, nlist_check_kernel)
__global__ void publish_only_kernel(const unsigned int* d_result, unsigned long long* d_published, unsigned int stamp)
    {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        __hip_atomic_store(d_published, ((unsigned long long)stamp << 32) | (unsigned long long)(*d_result), __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }

// STEP_TWO false: the pass starts at the first half step (accelerations read from one.d_accel), for integration
// methods whose two half steps cannot be joined (a thermostat needs the kinetic energy in between).
template<bool CHECK, bool ORTHO, bool STEP_TWO = true> __global__ void __launch_bounds__(256) fused_step_kernel(hmd_fused_step_args_t a)
    {
    const unsigned int g = blockIdx.x * blockDim.x + threadIdx.x;
    bool moved = false;
    if (g < a.one.group_size)
        {
        const unsigned int idx = a.one.d_group_members ? a.one.d_group_members[g] : g;
        double4 f = make_double4(0, 0, 0, 0);
        if (STEP_TWO)
            {
            f = hmd_ld4(a.two.d_net_force + idx);
            if (a.two.d_net_force2)
                {
                const double4 f2 = hmd_ld4(a.two.d_net_force2 + idx);
                f.x = HMD_ADD(f.x, f2.x);
                f.y = HMD_ADD(f.y, f2.y);
                f.z = HMD_ADD(f.z, f2.z);
                }
            }
        moved = hmd_fused_step_particle<ORTHO, STEP_TWO>(a, idx, f, CHECK);
        }
    if (CHECK)
        {
        if (__ballot(moved) && (threadIdx.x & 63) == 0)
            atomicMax(a.check.d_result, a.check.checkn);
        }
    }

__global__ void __launch_bounds__(256) accel_kernel(hmd_scalar3* d_accel, const hmd_scalar4* d_vel, const hmd_scalar4* d_nf, unsigned int N)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N)
        return;
    const double4 f = hmd_ld4(d_nf + idx);
    const double minv = 1.0 / d_vel[idx].w;
    hmd_scalar3 acc;
    acc.x = HMD_MUL(f.x, minv);
    acc.y = HMD_MUL(f.y, minv);
    acc.z = HMD_MUL(f.z, minv);
    d_accel[idx] = acc;
    }

template<bool VIRIAL> __global__ void __launch_bounds__(256) sum_net_force_kernel(hmd_sum_net_force_args_t a)
    {
    const unsigned int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.nparticles)
        return;
    double4 nf = make_double4(0, 0, 0, 0), nt = make_double4(0, 0, 0, 0);
    double nv[6] = {0, 0, 0, 0, 0, 0};
    if (!a.clear)
        {
        nf = hmd_ld4(a.d_net_force + idx);
        if (a.d_net_torque)
            nt = hmd_ld4(a.d_net_torque + idx);
        if (VIRIAL)
            for (int c = 0; c < 6; c++)
                nv[c] = a.d_net_virial[c * a.net_virial_pitch + idx];
        }
#pragma unroll
    for (int k = 0; k < 6; k++)
        {
        if (a.force_list.f[k])
            {
            const double4 f = hmd_ld4(a.force_list.f[k] + idx);
            nf.x = HMD_ADD(nf.x, f.x);
            nf.y = HMD_ADD(nf.y, f.y);
            nf.z = HMD_ADD(nf.z, f.z);
            nf.w = HMD_ADD(nf.w, f.w);
            if (VIRIAL && a.force_list.v[k])
                for (int c = 0; c < 6; c++)
                    nv[c] = HMD_ADD(nv[c], a.force_list.v[k][c * a.force_list.vpitch[k] + idx]);
            if (a.d_net_torque && a.force_list.t[k])
                {
                const double4 t = hmd_ld4(a.force_list.t[k] + idx);
                nt.x += t.x;
                nt.y += t.y;
                nt.z += t.z;
                nt.w += t.w;
                }
            }
        }
    hmd_st4(a.d_net_force + idx, nf.x, nf.y, nf.z, nf.w);
    if (a.d_net_torque)
        hmd_st4(a.d_net_torque + idx, nt.x, nt.y, nt.z, nt.w);
    if (VIRIAL)
        for (int c = 0; c < 6; c++)
            a.d_net_virial[c * a.net_virial_pitch + idx] = nv[c];
    }

// ---- thermo: fixed-shape two-level tree so the sums are reproducible run to run ----
constexpr unsigned int TH_BLOCKS = 1024;
constexpr unsigned int TH_BLOCK = 256;
constexpr unsigned int TH_VALS = 6;

__device__ __forceinline__ void block_reduce6(double (&v)[TH_VALS], double* s_part)
    {
    const unsigned int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (unsigned int q = 0; q < TH_VALS; q++)
        {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1)
            v[q] += __shfl_xor(v[q], m, 64);
        if (lane == 0)
            s_part[wave * TH_VALS + q] = v[q];
        }
    __syncthreads();
    if (threadIdx.x == 0)
        for (unsigned int q = 0; q < TH_VALS; q++)
            {
            double t = 0.0;
            for (unsigned int w = 0; w < TH_BLOCK / 64; w++)
                t += s_part[w * TH_VALS + q];
            v[q] = t;
            }
    }

__global__ void __launch_bounds__(TH_BLOCK) thermo_partial_kernel(hmd_thermo_args_t a)
    {
    __shared__ double s_part[(TH_BLOCK / 64) * TH_VALS];
    double v[TH_VALS] = {0, 0, 0, 0, 0, 0};
    for (unsigned int g = blockIdx.x * TH_BLOCK + threadIdx.x; g < a.group_size; g += TH_BLOCKS * TH_BLOCK)
        {
        const unsigned int idx = a.d_group_members ? a.d_group_members[g] : g;
        const double4 vel = hmd_ld4(a.d_vel + idx);
        v[0] += 0.5 * vel.w * (vel.x * vel.x + vel.y * vel.y + vel.z * vel.z);
        v[1] += a.d_net_force[idx].w;
        if (a.d_net_virial)
            v[2] += a.d_net_virial[0 * a.net_virial_pitch + idx] + a.d_net_virial[3 * a.net_virial_pitch + idx]
                    + a.d_net_virial[5 * a.net_virial_pitch + idx];
        v[3] += vel.w * vel.x;
        v[4] += vel.w * vel.y;
        v[5] += vel.w * vel.z;
        }
    block_reduce6(v, s_part);
    if (threadIdx.x == 0)
        {
        double* out = reinterpret_cast<double*>(a.d_scratch) + blockIdx.x * TH_VALS;
        for (unsigned int q = 0; q < TH_VALS; q++)
            out[q] = v[q];
        }
    }

// the second half step and the partial sums of thermo_partial_kernel (same particles per thread, same order) in one pass
__global__ void __launch_bounds__(TH_BLOCK) step_two_thermo_kernel(hmd_step_two_args_t s, hmd_thermo_args_t a)
    {
    __shared__ double s_part[(TH_BLOCK / 64) * TH_VALS];
    double v[TH_VALS] = {0, 0, 0, 0, 0, 0};
    for (unsigned int g = blockIdx.x * TH_BLOCK + threadIdx.x; g < a.group_size; g += TH_BLOCKS * TH_BLOCK)
        {
        const unsigned int idx = a.d_group_members ? a.d_group_members[g] : g;
        double pe;
        const double4 vel = step_two_particle(s, idx, pe);
        v[0] += 0.5 * vel.w * (vel.x * vel.x + vel.y * vel.y + vel.z * vel.z);
        v[1] += pe;
        v[3] += vel.w * vel.x;
        v[4] += vel.w * vel.y;
        v[5] += vel.w * vel.z;
        }
    block_reduce6(v, s_part);
    if (threadIdx.x == 0)
        {
        double* out = reinterpret_cast<double*>(a.d_scratch) + blockIdx.x * TH_VALS;
        for (unsigned int q = 0; q < TH_VALS; q++)
            out[q] = v[q];
        }
    }

__global__ void __launch_bounds__(TH_BLOCK) thermo_final_kernel(hmd_thermo_args_t a)
    {
    __shared__ double s_part[(TH_BLOCK / 64) * TH_VALS];
    double v[TH_VALS] = {0, 0, 0, 0, 0, 0};
    const double* in = reinterpret_cast<const double*>(a.d_scratch);
    for (unsigned int b = threadIdx.x; b < TH_BLOCKS; b += TH_BLOCK)
        for (unsigned int q = 0; q < TH_VALS; q++)
            v[q] += in[b * TH_VALS + q];
    block_reduce6(v, s_part);
    if (threadIdx.x == 0)
        {
        for (unsigned int q = 0; q < TH_VALS; q++)
            a.d_out[q] = v[q];
        a.d_out[6] = 0.0;
        a.d_out[7] = 0.0;
        if (a.d_out_host)
            for (unsigned int q = 0; q < 8; q++)
                __hip_atomic_store(a.d_out_host + q, q < TH_VALS ? v[q] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    } // namespace

extern "C" int hmd_nvt_rescale_step_one(const hmd_step_one_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nvt_rescale_step_one: args is NULL");
    HMD_REQUIRE(args->d_pos && args->d_vel && args->d_accel && args->d_image, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nvt_rescale_step_one: pos, vel, accel and image are required");
    if (args->group_size == 0)
        return HMD_OK;
    hipLaunchKernelGGL(step_one_kernel, dim3(hmd_div_up(args->group_size, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_nvt_rescale_step_two(const hmd_step_two_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nvt_rescale_step_two: args is NULL");
    HMD_REQUIRE(args->d_vel && args->d_accel && args->d_net_force, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nvt_rescale_step_two: vel, accel and net_force are required");
    if (args->group_size == 0)
        return HMD_OK;
    hipLaunchKernelGGL(step_two_kernel, dim3(hmd_div_up(args->group_size, 256)), dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_nve_step_two_one_check(const hmd_fused_step_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nve_step_two_one_check: args is NULL");
    const hmd_fused_step_args_t& a = *args;
    HMD_REQUIRE(a.two.d_vel && a.two.d_net_force && a.one.d_pos && a.one.d_vel && a.one.d_image,
                HMD_ERR_INVALID_ARGUMENT, "hmd_nve_step_two_one_check: pos, vel, accel, image and net_force are required");
    HMD_REQUIRE(a.two.d_vel == a.one.d_vel && a.two.d_accel == a.one.d_accel && a.two.d_group_members == a.one.d_group_members
                    && a.two.group_size == a.one.group_size,
                HMD_ERR_INVALID_ARGUMENT, "hmd_nve_step_two_one_check: both halves must act on the same arrays and group");
    HMD_REQUIRE((a.d_cpos == nullptr) == (a.d_slot_of == nullptr), HMD_ERR_INVALID_ARGUMENT,
                "hmd_nve_step_two_one_check: d_cpos and d_slot_of go together");
    HMD_REQUIRE(!a.one.use_limit, HMD_ERR_UNSUPPORTED, "hmd_nve_step_two_one_check: the displacement limit needs the separate step-one driver");
    const bool check = a.check.d_result != nullptr;
    if (check)
        {
        HMD_REQUIRE(a.check.d_last_pos && a.check.d_rcut_max, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_nve_step_two_one_check: the check needs last_pos and rcut_max");
        HMD_REQUIRE(a.one.d_group_members == nullptr && a.one.group_size == a.check.N, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_nve_step_two_one_check: the check needs the group of all particles");
        }
    if (a.one.group_size == 0)
        {
        if (check && a.check.d_published)
            hipLaunchKernelGGL(publish_only_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a.check.d_result, a.check.d_published, a.check.stamp);
        return hmd_launch_status();
        }
    const dim3 grid(hmd_div_up(a.one.group_size, 256));
    if (!check)
        hipLaunchKernelGGL((fused_step_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (hmd_box_is_ortho(a.check.box))
        hipLaunchKernelGGL((fused_step_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((fused_step_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    // a separate one-wave kernel hands the flag to the host: making every workgroup's atomic visible to a
    // "last workgroup" inside the big kernel would need a device-wide fence per workgroup, which costs
    // far more across the eight XCDs than this launch
    if (check && a.check.d_published)
        hipLaunchKernelGGL(publish_only_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a.check.d_result, a.check.d_published, a.check.stamp);
    return hmd_launch_status();
    }

extern "C" int hmd_nvt_rescale_step_one_check(const hmd_fused_step_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_nvt_rescale_step_one_check: args is NULL");
    const hmd_fused_step_args_t& a = *args;
    HMD_REQUIRE(a.one.d_pos && a.one.d_vel && a.one.d_accel && a.one.d_image, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nvt_rescale_step_one_check: pos, vel, accel and image are required");
    HMD_REQUIRE((a.d_cpos == nullptr) == (a.d_slot_of == nullptr), HMD_ERR_INVALID_ARGUMENT,
                "hmd_nvt_rescale_step_one_check: d_cpos and d_slot_of go together");
    HMD_REQUIRE(!a.one.use_limit, HMD_ERR_UNSUPPORTED, "hmd_nvt_rescale_step_one_check: the displacement limit needs the separate step-one driver");
    const bool check = a.check.d_result != nullptr;
    if (check)
        {
        HMD_REQUIRE(a.check.d_last_pos && a.check.d_rcut_max, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_nvt_rescale_step_one_check: the check needs last_pos and rcut_max");
        HMD_REQUIRE(a.one.d_group_members == nullptr && a.one.group_size == a.check.N, HMD_ERR_INVALID_ARGUMENT,
                    "hmd_nvt_rescale_step_one_check: the check needs the group of all particles");
        }
    if (a.one.group_size != 0)
        {
        const dim3 grid(hmd_div_up(a.one.group_size, 256));
        if (!check)
            hipLaunchKernelGGL((fused_step_kernel<false, true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else if (hmd_box_is_ortho(a.check.box))
            hipLaunchKernelGGL((fused_step_kernel<true, true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL((fused_step_kernel<true, false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
        }
    if (check && a.check.d_published)
        hipLaunchKernelGGL(publish_only_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a.check.d_result, a.check.d_published, a.check.stamp);
    return hmd_launch_status();
    }

extern "C" int hmd_compute_accelerations(hmd_scalar3* d_accel, const hmd_scalar4* d_vel, const hmd_scalar4* d_net_force, unsigned int N, void* stream)
    {
    HMD_REQUIRE(d_accel && d_vel && d_net_force, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_accelerations: all arrays are required");
    if (N == 0)
        return HMD_OK;
    hipLaunchKernelGGL(accel_kernel, dim3(hmd_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, d_accel, d_vel, d_net_force, N);
    return hmd_launch_status();
    }

extern "C" int hmd_integrator_sum_net_force(const hmd_sum_net_force_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_integrator_sum_net_force: args is NULL");
    HMD_REQUIRE(args->d_net_force, HMD_ERR_INVALID_ARGUMENT, "hmd_integrator_sum_net_force: d_net_force is required");
    HMD_REQUIRE(!args->compute_virial || args->d_net_virial, HMD_ERR_INVALID_ARGUMENT,
                "hmd_integrator_sum_net_force: compute_virial needs d_net_virial");
    if (args->nparticles == 0)
        return HMD_OK;
    const dim3 grid(hmd_div_up(args->nparticles, 256));
    if (args->compute_virial)
        hipLaunchKernelGGL((sum_net_force_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, *args);
    else
        hipLaunchKernelGGL((sum_net_force_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" size_t hmd_thermo_scratch_bytes(void)
    {
    return sizeof(double) * TH_BLOCKS * TH_VALS;
    }

extern "C" int hmd_compute_thermo(const hmd_thermo_args_t* args, void* stream)
    {
    HMD_REQUIRE(args, HMD_ERR_INVALID_ARGUMENT, "hmd_compute_thermo: args is NULL");
    HMD_REQUIRE(args->d_out && args->d_scratch && args->d_vel && args->d_net_force, HMD_ERR_INVALID_ARGUMENT,
                "hmd_compute_thermo: out, scratch, vel and net_force are required");
    hipLaunchKernelGGL(thermo_partial_kernel, dim3(TH_BLOCKS), dim3(TH_BLOCK), 0, (hipStream_t)stream, *args);
    hipLaunchKernelGGL(thermo_final_kernel, dim3(1), dim3(TH_BLOCK), 0, (hipStream_t)stream, *args);
    return hmd_launch_status();
    }

extern "C" int hmd_nvt_rescale_step_two_thermo(const hmd_step_two_args_t* two, const hmd_thermo_args_t* thermo, void* stream)
    {
    HMD_REQUIRE(two && thermo, HMD_ERR_INVALID_ARGUMENT, "hmd_nvt_rescale_step_two_thermo: two or thermo is NULL");
    HMD_REQUIRE(two->d_vel && two->d_accel && two->d_net_force, HMD_ERR_INVALID_ARGUMENT,
                "hmd_nvt_rescale_step_two_thermo: vel, accel and net_force are required");
    HMD_REQUIRE(thermo->d_out && thermo->d_scratch, HMD_ERR_INVALID_ARGUMENT, "hmd_nvt_rescale_step_two_thermo: out and scratch are required");
    HMD_REQUIRE(thermo->d_vel == two->d_vel && thermo->d_net_force == two->d_net_force && thermo->d_group_members == two->d_group_members
                    && thermo->group_size == two->group_size,
                HMD_ERR_UNSUPPORTED, "hmd_nvt_rescale_step_two_thermo: the sums must be over the group, velocities and net force of the half step");
    HMD_REQUIRE(!thermo->d_net_virial && !two->d_net_force2, HMD_ERR_UNSUPPORTED,
                "hmd_nvt_rescale_step_two_thermo: virial sums and a second force array need the separate drivers");
    hipLaunchKernelGGL(step_two_thermo_kernel, dim3(TH_BLOCKS), dim3(TH_BLOCK), 0, (hipStream_t)stream, *two, *thermo);
    hipLaunchKernelGGL(thermo_final_kernel, dim3(1), dim3(TH_BLOCK), 0, (hipStream_t)stream, *thermo);
    return hmd_launch_status();
    }
