// step_body.cuh -- one particle's share of the joined integration pass: second half step of time step t, first half
// step of t + 1, the neighbor list's distance check of the new position and the slot-ordered position copy
// (hoomd/md/TwoStepConstantVolumeGPU.cu:215-260 and :102-155, hoomd/md/NeighborListGPU.cu:42-84, arithmetic of the CPU
// class hoomd/md/TwoStepConstantVolume.cc:31-79,233-261).  Used by the integration kernel and by the pair-force tile
// kernel when the pass rides on it; every product and sum is rounded on its own in both (the pragma below holds for
// this function whatever the including unit's setting is), so the two give the same bits.
#ifndef HMD_STEP_BODY_CUH
#define HMD_STEP_BODY_CUH

#include "common.cuh"

// f: the net force on the particle (w unused).  STEP_TWO false: the pass starts at the first half step with the
// acceleration read from one.d_accel.  Returns whether the particle moved far enough for the neighbor list to be stale.
template<bool ORTHO, bool STEP_TWO>
__device__ __forceinline__ bool hmd_fused_step_particle(const hmd_fused_step_args_t& a, unsigned int idx, const double4& f, bool check)
    {
#pragma clang fp contract(off)
    const double4 v = hmd_ld4(a.one.d_vel + idx);
    const double4 p = hmd_ld4(a.one.d_pos + idx);
    // (everything the pass reads is asked for here, ahead of the stores: the compiler may not move a load across a store
    // through another pointer, and read where they are used the position of the last build and the slot cost every wave a
    // second round trip to memory)
    double4 last = make_double4(0, 0, 0, 0);
    if (check)
        last = hmd_ld4(a.check.d_last_pos + idx);
    unsigned int slot = 0;
    if (a.d_cpos)
        slot = a.d_slot_of[idx];
    // (the cutoff of the particle's type likewise: with one type a uniform read, otherwise asked for as soon as the type is known)
    double rmin = 0.0;
    if (check)
        rmin = a.check.d_rcut_max[a.check.ntypes == 1u ? 0u : hmd_double_as_uint(p.w)];
    hmd_scalar3 acc;
    double vx = v.x, vy = v.y, vz = v.z;
    if (STEP_TWO)
        {
        // ---- step two (t) ----
        const double minv = 1.0 / v.w;
        acc.x = HMD_MUL(f.x, minv);
        acc.y = HMD_MUL(f.y, minv);
        acc.z = HMD_MUL(f.z, minv);
        const double hdt = HMD_MUL(0.5, a.two.deltaT);
        vx = HMD_ADD(HMD_MUL(v.x, a.two.rescale_factor), HMD_MUL(hdt, acc.x));
        vy = HMD_ADD(HMD_MUL(v.y, a.two.rescale_factor), HMD_MUL(hdt, acc.y));
        vz = HMD_ADD(HMD_MUL(v.z, a.two.rescale_factor), HMD_MUL(hdt, acc.z));
        if (a.two.d_accel)
            a.two.d_accel[idx] = acc;
        }
    else
        acc = a.one.d_accel[idx];
    // ---- step one (t+1) ----
    const double rs = hmd_rescale_factor(a.one);
    vx = HMD_MUL(HMD_ADD(vx, HMD_MUL(HMD_MUL(0.5, acc.x), a.one.deltaT)), rs);
    vy = HMD_MUL(HMD_ADD(vy, HMD_MUL(HMD_MUL(0.5, acc.y), a.one.deltaT)), rs);
    vz = HMD_MUL(HMD_ADD(vz, HMD_MUL(HMD_MUL(0.5, acc.z), a.one.deltaT)), rs);
    double x = HMD_ADD(p.x, HMD_MUL(a.one.deltaT, vx));
    double y = HMD_ADD(p.y, HMD_MUL(a.one.deltaT, vy));
    double z = HMD_ADD(p.z, HMD_MUL(a.one.deltaT, vz));
    // few particles cross the box boundary in a step: touch the image only for those
    hmd_int3 dimg = {0, 0, 0};
    hmd_wrap(a.one.box, x, y, z, dimg.x, dimg.y, dimg.z);
    hmd_st4(a.one.d_pos + idx, x, y, z, p.w);
    hmd_st4(a.one.d_vel + idx, vx, vy, vz, v.w);
    if (dimg.x | dimg.y | dimg.z)
        {
        hmd_int3 img = a.one.d_image[idx];
        img.x += dimg.x;
        img.y += dimg.y;
        img.z += dimg.z;
        a.one.d_image[idx] = img;
        }
    if (a.d_cpos)
        hmd_st4(a.d_cpos + slot, x, y, z, p.w);
    // ---- distance check of the new position (hoomd/md/NeighborListGPU.cu:42-84) ----
    bool moved = false;
    if (check)
        {
        double dx = HMD_SUB(x, HMD_MUL(a.check.lambda.x, last.x));
        double dy = HMD_SUB(y, HMD_MUL(a.check.lambda.y, last.y));
        double dz = HMD_SUB(z, HMD_MUL(a.check.lambda.z, last.z));
        hmd_min_image<ORTHO>(a.check.box, dx, dy, dz);
        const double rmax = HMD_ADD(rmin, a.check.r_buff);
        const double delta_max = HMD_SUB(HMD_MUL(rmax, a.check.lambda_min), rmin) / 2.0;
        const double maxsq = (delta_max > 0) ? HMD_MUL(delta_max, delta_max) : 0.0;
        moved = HMD_ADD(HMD_ADD(HMD_MUL(dx, dx), HMD_MUL(dy, dy)), HMD_MUL(dz, dz)) >= maxsq;
        }
    return moved;
    }

#endif
