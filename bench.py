#!/usr/bin/env python3
"""Headline benchmark: LJ liquid, N = 1,048,576 particles per GPU (rho = 0.84, kT = 1.5, r_cut = 2.5,
buffer = 0.4, dt = 0.005, fp64), cell list -> neighbor list -> LJ pair force -> velocity Verlet.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --scaling strong                   (configuration C: 256x256x128 sites over 8 GPUs)
    python bench.py --gpus 2 --backend gloo --lattice 48x48x48  (rehearsal: ranks sharing one GPU, host-staged)

One "step" is one MD time step of the whole system.  Every run first equilibrates, untimed and whatever W
is: the lattice is melted under a Berendsen thermostat and switched to NVE for the last 200 of --equil
steps (config.equilibration_steps).  Then W untimed warm-up steps, then exactly K NVE steps are timed between
barrier + device synchronisation on both sides; the maximum over ranks is used.  With N > 1 GPUs the system is split
by spatial domain decomposition with a halo exchange over RCCL: --scaling weak (default) gives every GPU a sub-box of
--lattice sites (128x128x64), --scaling strong splits ONE box of --lattice sites (256x256x128) over the GPUs.
Rank 0 prints one JSON line.

Started with --gpus N > 1 and no WORLD_SIZE in the environment, this file is its own launcher: the parent process
touches no GPU, starts N fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), relays
rank 0's result line and exits non-zero as soon as any child has failed (the others are then stopped) or when
--launch-timeout seconds have passed.

`value` is particle-steps per second of the whole job; `timesteps_per_s` rides along.
`roofline` prices the pair-force kernel: algorithmic bytes per launch = 4 * (listed neighbors) +
76 * N, divided by the kernel's average duration measured with HIP events on the launch stream
inside the timed region; `fp64_tflops` is 37 flop per listed pair (minimum image excluded: the tile path has
none) over the same duration, next to the 78.6 TFLOP/s fp64 vector peak, so that an ALU-bound kernel shows.
`cpu_baseline` times the CPU oracle (a C restatement of the reference's CPU path: half neighbor list + Newton's
third law) as ONE domain-decomposed run over the host cores of this process's CPU share, started from the state the
GPU run ended in, for a bounded number of steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector (non-matrix) peak
FLOP_PER_LISTED_PAIR = 37.0  # dx (3), r^2 (5), cutoff test, LJ evaluator (~14 + 1 divide), 3 force FMAs, energy


def grid_for(n):
    """Domain grid for n ranks: the split that keeps the per-GPU sub-box at 128 x 128 x 64 sites."""
    return {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(n) or (n, 1, 1)


def cpu_baseline(snapshot, steps=150):
    """The reference's CPU mode is one MPI rank per core, each with its own sub-box and ghost layer.  Here: the
    oracle's restatement of that CPU path (binned half list with ghosts, third-law pair loop, velocity Verlet,
    distance check) on one sub-box per core of this process's CPU share (at most 15), halos / migration / rebuild vote
    over gloo (scripts/cpu_decomposed.py), started from the state the GPU run ended in -- same N, same liquid --
    for a bounded number of steps."""
    import subprocess
    import tempfile
    import numpy as np
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    # (15, not 16: this process keeps the GPU, and a GPU box admits 16 processes of one job in all)
    cores = max(1, min(15, share))
    L = np.asarray(snapshot.configuration.box[:3], dtype=np.float64)
    pos = np.asarray(snapshot.particles.position, dtype=np.float64)
    pos = pos - np.floor((pos + L / 2) / L) * L
    with tempfile.TemporaryDirectory() as tmp:
        state, out = os.path.join(tmp, "state.npz"), os.path.join(tmp, "out.json")
        np.savez(state, position=pos, velocity=np.asarray(snapshot.particles.velocity, dtype=np.float64), L=L)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        script = os.path.join(ROOT, "scripts", "cpu_decomposed.py")
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, script, str(r), str(cores), str(port), state, str(steps), out], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.PIPE, env=env) for r in range(cores)]
        errs = [p.communicate(timeout=900)[1] for p in procs]
        wall = time.perf_counter() - t0
        bad = [r for r, p in enumerate(procs) if p.returncode != 0]
        if bad:
            raise RuntimeError(f"CPU baseline rank {bad[0]} failed: {errs[bad[0]].decode(errors='replace')[-2000:]}")
        with open(out) as f:
            res = json.load(f)
    N = res["N"]
    return {
        "value": N * res["steps"] / res["seconds"], "unit": "particle-steps/s", "cores": cores, "kind": "port",
        "launcher": f"{cores} processes (one per core, OMP_NUM_THREADS=1), torch.distributed gloo over 127.0.0.1, scripts/cpu_decomposed.py",
        "sample": f"ONE decomposed run of the same system from the state the GPU run ended in: N={N}, domain grid {res['grid']}, "
                  f"{res['steps']} NVE steps in {res['seconds']:.1f} s ({wall:.1f} s with start-up), {res['rebuilds']} neighbor-list rebuilds with migration + ghost "
                  f"exchange, ghost positions every other step, {res['mean_ghosts_per_rank']:.0f} ghosts per rank; half list + third law per rank",
        "timesteps_per_s": res["steps"] / res["seconds"],
        "total_energy_per_particle_at_end": (res["kinetic_energy"] + res["potential_energy"]) / N,
    }


def launch_ranks(args, argv):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: starts one fresh child process per rank
    and never touches a GPU itself (children are new processes, nothing is re-executed in a process that has
    initialised the device).  Role of `mpirun -n 8` in the reference's own distributed tests
    (hoomd/md/test/CMakeLists.txt:22-32,66-75)."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    if args.backend == "nccl":
        import torch  # (counting devices does not initialise the GPU)
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} over RCCL needs {n} GPUs, one rank per GPU; this machine shows {have}.  "
                  f"(Ranks sharing a GPU can only be rehearsed with --backend gloo.)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HOOMD_AMD_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), start_new_session=True))

    def stop_all():
        for p in procs:  # exactly the process groups started above
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                p.wait()

    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().decode(errors="replace").splitlines()), daemon=True)
    reader.start()
    deadline = time.time() + args.launch_timeout
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print(f"bench.py: rank {bad[0][0]} exited with status {bad[0][1]}; stopping the other ranks", file=sys.stderr)
                rc = 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                print(f"bench.py: ranks still running after --launch-timeout {args.launch_timeout} s; stopping them", file=sys.stderr)
                rc = 3
                break
            time.sleep(0.1)
    finally:
        stop_all()
    reader.join(timeout=5)
    result = [ln for ln in lines if ln.startswith("{")]
    if rc == 0 and not result:
        print("bench.py: rank 0 ended without a result line", file=sys.stderr)
        rc = 1
    if rc == 0:
        print(result[-1], flush=True)
    return rc


def main():
    # Libraries print banners on the C-level standard output (RCCL does at start-up): everything written to
    # file descriptor 1 goes to standard error until the result line, which is then the only line on stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: the protocol of the measurement plan -- 10,000 steps of equilibration, 2,000 of warm-up, 5,000 timed)
    ap.add_argument("--steps", type=int, default=5000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--equil", type=int, default=10000,
                    help="untimed equilibration steps before the warm-up: the lattice melts under the thermostat, the last 200 are NVE")
    ap.add_argument("--scaling", type=str, default="weak", choices=["weak", "strong"],
                    help="weak: --lattice sites PER GPU (default 128x128x64, N = 2^20 per GPU); strong: --lattice sites IN ALL "
                         "(default 256x256x128, N = 2^23: configuration C) split over the GPUs by the least-surface grid")
    ap.add_argument("--lattice", type=str, default=None, help="lattice sites per GPU (weak scaling) or of the whole system (strong scaling)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds after which a self-launched multi-rank run is stopped")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tpp", type=int, default=0)
    ap.add_argument("--workload", type=str, default="lj", choices=["lj", "ka", "kg"],
                    help="lj: the headline LJ liquid; side measurements: ka = Kob-Andersen 80:20 mixture at rho 1.2 under a Bussi "
                         "thermostat, kg = Kremer-Grest bead-spring melt (WCA + FENE bonds, chains of 64 along the lattice's x rows at rho = 0.85)")
    ap.add_argument("--time-all", action="store_true", help="time every kernel family with HIP events (perturbs the run) and print the table to stderr")
    ap.add_argument("--time-every", type=int, default=4,
                    help="bracket one pair-force launch in this many with HIP events for the roofline (1: every launch)")
    ap.add_argument("--cprofile", action="store_true", help="profile the host side of the timed region (rank 0, to stderr)")
    ap.add_argument("--force-dd", action="store_true", help="run the domain-decomposition code path even on one rank (rehearsal of the RCCL calls)")
    ap.add_argument("--cpu-steps", type=int, default=150, help="steps of the decomposed CPU baseline run (about 0.1 s each on 16 cores at N = 2^20)")
    ap.add_argument("--rider", action="store_true",
                    help="run the half steps inside the force kernel (one kernel per step) instead of as a kernel of their own; off by default: "
                         "it buys under one per cent of the step, and the roofline then prices a kernel that is harder to read")
    ap.add_argument("--no-rider", action="store_true", help="(the default; kept for old command lines)")
    ap.add_argument("--no-tiles", action="store_true", help="force the neighbor list to index rows (the tile path is chosen by cell occupancy otherwise)")
    ap.add_argument("--no-overlap", action="store_true", help="domain decomposition: do not compute interior blocks during the halo exchange")
    ap.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse ranks sharing one GPU)")
    args = ap.parse_args()
    if args.lattice is None:
        args.lattice = "128x128x64" if args.scaling == "weak" else "256x256x128"
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only starts the ranks and relays the result
        os.dup2(real_stdout, 1)
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.backend == "nccl" and world > max(torch.cuda.device_count(), 1):
        raise SystemExit(f"--gpus {world} over RCCL needs {world} GPUs, one rank per GPU; this machine shows {torch.cuda.device_count()} "
                         "(ranks sharing a GPU can only be rehearsed with --backend gloo)")

    import hoomd_amd as hoomd
    from hoomd_amd import _hoomd_amd as cpp
    import lj_liquid

    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    decomposition = None
    lat = tuple(int(x) for x in args.lattice.split("x"))
    if args.scaling == "weak":
        grid = grid_for(world)
        n, per_gpu = tuple(a * g for a, g in zip(lat, grid)), lat
    else:
        # one box for all GPUs, split by the least inter-domain surface (hoomd/DomainDecomposition.cc:327-408)
        from hoomd_amd.comm import find_grid
        grid = find_grid(world, [float(a) for a in lat])
        n, per_gpu = lat, tuple(a // g for a, g in zip(lat, grid))
    if world > 1 or args.force_dd:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # HOOMD_AMD_COMM_PRIORITY=high: exchange stream (and RCCL's own stream when torch issues the collectives) on a
        # high-priority hardware queue, so their kernels run beside the force kernel of the default stream instead of
        # lining up behind it.  Normal priority is the default: no run with peers at high priority is on record.
        if os.environ.get("HOOMD_AMD_COMM_PRIORITY") == "high":
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend=args.backend)
        from hoomd_amd import comm
        decomposition = comm.DomainDecomposition(grid=grid)

    if args.workload == "ka":
        thermostat = hoomd.md.methods.thermostats.Bussi(kT=1.0)
        sim, cell, lj, method = lj_liquid.make_ka_simulation(hoomd, n, thermostat=thermostat, decomposition=decomposition)
    elif args.workload == "kg":
        thermostat = hoomd.md.methods.thermostats.Berendsen(kT=1.0, tau=0.5)
        # chains of 64 at rho = 0.85 (0.97 along the chains, 1.1013 across them on the start lattice)
        sim, cell, lj, fene, method = lj_liquid.make_kg_simulation(hoomd, n, chain_length=64, across=(1.0 / (0.85 * 0.97)) ** 0.5, thermostat=thermostat,
                                                                   decomposition=decomposition)
    else:
        thermostat = hoomd.md.methods.thermostats.Berendsen(kT=1.5, tau=0.5)
        sim, cell, lj, method = lj_liquid.make_simulation(hoomd, n, thermostat=thermostat, decomposition=decomposition)
    N_total = sim.state.N_particles
    if args.no_overlap:
        sim.run(0)
        cell._cpp_obj.overlap = False
    if args.no_tiles:
        cell._attach(sim) if cell._cpp_obj is None else None
        cell._cpp_obj.tiles = False
    if args.tpp:
        sim.run(0)
        lj._cpp_obj.threads_per_particle = args.tpp
    sim.run(0)
    sim.operations.integrator._cpp_obj.step_rider = bool(args.rider) and not args.no_rider

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- equilibration (always, untimed): melt the lattice under the thermostat, then NVE; every buffer that grows
    # with the liquid's neighbor counts (rows, cell slots, tile capacity) reaches its size here ----
    melt = max(args.equil - 200, 0)
    sim.run(melt)
    if args.workload != "ka":
        method.thermostat = None  # the Kob-Andersen case keeps its thermostat, as the reference configuration does
    cell._cpp_obj.resetStats()
    sim.run(args.equil - melt)
    # ---- warm-up: W more steps of exactly what is timed ----
    sim.run(args.warmup)
    # mean rebuild period of the liquid as it is timed, from the steps just run (the timed window may hold one rebuild
    # more or fewer than steps / period: with 20 steps and a period of 6 that is 3 or 4, a difference of 10 % of the line)
    steps_before = args.equil - melt + args.warmup
    builds_before = int(cell.num_builds)
    mean_period = steps_before / builds_before if builds_before else None

    # ---- timed region ----
    cell._cpp_obj.resetStats()
    cpp.reset_kernel_timing()
    # launches predicated off by a stale neighbor list (one per rebuild step) end after their first loads: samples
    # under half the median of their kernel are not counted
    if args.time_all:
        cpp.enable_kernel_timing(True, "", 0.0)
    else:
        # one launch in args.time_every is bracketed by events (a pair of events costs the stream a few microseconds)
        cpp.enable_kernel_timing(True, "pair_force", -500.0, max(1, args.time_every))
    barrier()
    prof = None
    if args.cprofile and rank == 0:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    cpp.poll_wait_seconds(True)
    t0 = time.perf_counter()
    sim.run(args.steps)
    host_wait = cpp.poll_wait_seconds(False)  # (before the barrier: the time the run loop itself waited for published words)
    barrier()
    elapsed = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
    timing = cpp.kernel_timing()
    cpp.enable_kernel_timing(False)
    if args.time_all and rank == 0:
        for k, (c, ms) in sorted(timing.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:24s} launches {c:7d}  total {ms:9.2f} ms  per step {1e3 * ms / args.steps:8.2f} us", file=sys.stderr)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # listed neighbors of this rank's particles (for the algorithmic byte count)
    pdata = sim._sysdef.getParticleData()
    n_local = pdata.N
    import numpy as np
    n_neigh = np.asarray(cell._cpp_obj.getNNeighArray(n_local), dtype=np.int64)
    listed = int(n_neigh.sum())
    calls, ms = timing.get("pair_force", (0, 0.0))
    # An NVE run with one pair force takes its half steps inside the force kernel ("pair_force_step": the kernel then
    # also moves the integration pass's 228 bytes per particle -- 180 + 124 + 64 for the three drivers it stands for,
    # less what their fusion saves); the roofline prices the kind of launch that makes up most of the timed steps.
    step_calls, step_ms = timing.get("pair_force_step", (0, 0.0))
    rider = step_calls > calls
    plain = {"launches": int(calls), "avg_launch_us": (1e3 * ms / calls) if calls else None}
    if rider:
        calls, ms = step_calls, step_ms
    # under domain decomposition one step's forces come from two launches (blocks that see no ghosts run
    # during the halo exchange, the rest after it): one step = one unit of work, its duration their sum
    avg_s = (ms / 1e3 / calls) if calls else float("nan")
    if "pair_force_interior" in timing and timing["pair_force_interior"][0]:
        ic, ims = timing["pair_force_interior"]
        avg_s = (ims / 1e3 / ic) + ((ms / 1e3 / calls) if calls else 0.0)
        calls = ic
    # the integration pass adds 196 bytes per particle (228 for the joined half steps + check + slot copy, less the 32 of
    # the force the same wave has just stored: reading it back is not compulsory traffic)
    INTEGRATION_BYTES = 196.0
    algo_bytes = 4.0 * listed + 76.0 * n_local + (INTEGRATION_BYTES * n_local if rider else 0.0)
    achieved = algo_bytes / avg_s / 1e9 if calls else float("nan")
    fp64_tflops = FLOP_PER_LISTED_PAIR * listed / avg_s / 1e12 if calls else float("nan")

    # HBM traffic of the pair-force kernel per launch from the L2 memory-side counters (FETCH_SIZE / WRITE_SIZE in
    # separate rocprofv3 --pmc passes, scripts/measure_traffic.sh): counters cannot be read from inside this
    # process, so the figure is CARRIED from the newest file in profiles/ measured on the same workload, N and kind of
    # launch; `traffic_source` names that file (null when none matches: other workload, other size, decomposed run)
    traffic, traffic_source = None, None
    kernel_key = "pair_force_tiles_kernel_step" if rider else "pair_force_tiles_kernel" if cell._cpp_obj.tiles_active else "pair_force_kernel"
    if world == 1 and decomposition is None:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            try:
                with open(path) as f:
                    tr = json.load(f)
                if tr.get("N") == n_local and tr.get("workload", "lj") == args.workload and kernel_key in tr["kernels"]:
                    traffic = tr["kernels"][kernel_key]["hbm_bytes_per_launch"]
                    traffic_source = "profiles/" + os.path.basename(path) + " (separate rocprofv3 --pmc run; carried, not measured by this process)"
                    break
            except (OSError, KeyError, ValueError):
                continue

    if rank == 0:
        out = {
            "metric": "particle-steps/s (whole job) of the LJ-liquid short-range MD step; timesteps/s alongside",
            "value": N_total * args.steps / elapsed,
            "unit": "particle-steps/s",
            "timesteps_per_s": args.steps / elapsed,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"Kob-Andersen 80:20 LJ mixture N={N_total} ({n[0]}x{n[1]}x{n[2]} lattice, {per_gpu[0]}x{per_gpu[1]}x{per_gpu[2]} per GPU) "
                             "rho=1.2 kT=1.0 r_cut=2.5 sigma_ij buffer=0.4 dt=0.005, Cell nlist + LJ + ConstantVolume with Bussi thermostat, fp64"
                             if args.workload == "ka" else
                             f"Kremer-Grest melt N={N_total} ({n[0]}x{n[1]}x{n[2]} lattice, chains of 64) rho=0.85 kT=1.0 WCA r_cut=2^(1/6) shifted + "
                             "FENE k=30 r0=1.5, bond exclusions, buffer=0.4 dt=0.005, Cell nlist + ConstantVolume (NVE), fp64"
                             if args.workload == "kg" else
                             f"LJ liquid N={N_total} ({n[0]}x{n[1]}x{n[2]} lattice, {per_gpu[0]}x{per_gpu[1]}x{per_gpu[2]} per GPU) rho=0.84 kT=1.5 "
                             "r_cut=2.5 buffer=0.4 dt=0.005, Cell nlist + LJ + ConstantVolume (NVE), fp64"),
                "N": N_total, "equilibration_steps": args.equil, "domain_grid": list(grid), "rebuilds_in_timed_region": int(cell.num_builds),
                "mean_rebuild_period_before_timed_region": mean_period,
                "expected_rebuilds_in_window": (args.steps / mean_period) if mean_period else None,
                # share of the timed region the host spent waiting for words the device publishes (rebuild flags, build
                # results): close to 1 when the device bounds the run, small when the host's own work does
                "host_wait_fraction": host_wait / elapsed,
                "shortest_rebuild": int(cell.shortest_rebuild), "tile_capacity": int(cell._cpp_obj.tile_capacity), "mean_listed_neighbors": listed / max(n_local, 1),
            },
            "roofline": {
                "kernel": ("pair_force_tiles_kernel<EvaluatorPairLJ, STEP> (rank 0): pair force + second half step + next first half step + distance check"
                           if rider else "pair_force_tiles_kernel<EvaluatorPairLJ> (rank 0)" if cell._cpp_obj.tiles_active
                           else "pair_force_kernel<EvaluatorPairLJ> (rank 0)"),
                # the nearer ceiling of the two this kernel is priced against: "hbm" when its share of the HBM peak is the
                # larger one, "fp64-valu" when its share of the fp64 vector peak is (achieved / peak stay the HBM figures)
                "bound": "hbm" if achieved / HBM_PEAK_GBS >= fp64_tflops / FP64_VECTOR_PEAK_TFLOPS else "fp64-valu",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source, "avg_launch_us": 1e6 * avg_s, "launches": int(calls),
                "launches_note": f"launches bracketed by HIP events on their stream: one in {max(1, args.time_every)} of the timed region's", "algorithmic_bytes_per_launch": algo_bytes,
                "fp64_tflops": fp64_tflops, "fp64_peak_tflops": FP64_VECTOR_PEAK_TFLOPS, "fp64_frac": fp64_tflops / FP64_VECTOR_PEAK_TFLOPS,
                "pair_force_bytes_per_launch": 4.0 * listed + 76.0 * n_local, "integration_bytes_per_launch": INTEGRATION_BYTES * n_local if rider else 0.0,
                "launches_without_the_integration_pass": plain,
            },
        }
        if decomposition is not None:
            out["config"]["decomposition"] = {
                "backend": dist.get_backend(), "transport": decomposition.communicator.transport_name, "exchange_stream_high_priority": bool(cell._cpp_obj.comm_stream_high_priority),
                "exchange_stream_kept_after_probe": int(cell._cpp_obj.comm_stream_probe),
                "steps_with_forces_overlapping_the_exchange": int(cell._cpp_obj.num_overlapped),
                "ghosts_rank0": int(pdata.n_ghosts), "halo_exchanges_carrying_the_vote": int(decomposition.communicator.n_vote_exchanges),
                # rebuild-step exchanges (migration, ghost layer) whose message sizes stayed on the device until after the
                # exchange, and exchanges of counts that preceded a message
                "exchanges_with_sizes_read_on_the_device": int(decomposition.communicator.n_device_count_exchanges),
                "exchanges_of_counts": int(decomposition.communicator.n_count_exchanges)}
        if world == 1 and not args.no_cpu_baseline:
            if args.workload == "lj":
                out["cpu_baseline"] = cpu_baseline(sim.state.get_snapshot(), args.cpu_steps)
            else:
                out["cpu_baseline"] = None  # the CPU port covers the headline workload
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if world > 1 or args.force_dd:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
