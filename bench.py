#!/usr/bin/env python3
"""bench.py -- the reference's headline workload on MI355X.

Workload (BASELINE.json metric "SCF iterations/sec per GPU; MBE-2 wall-time @64 frags", configs[2]): the (H2O)64
cluster, MBE level 2, RHF/cc-pVDZ, exact four-centre ERIs held in HBM (Schwarz-screened at 1e-12 by default -- the
north star's wavefront-level screening; --schwarz-tol 0 forms every quartet), no distance cutoff: 64 monomer + 2016
dimer SCFs = 2080 fragments, GWH guess, e_tol 1e-10 / d_tol 1e-8.  One "step" = one complete
MBE-2 energy evaluation: every owned fragment through the engine (int1e -> ERI -> SCF to convergence with the
reference's dE / rms(dD) test and final rebuild), then ONE all-reduce of the zero-padded fragment-energy vector over
RCCL and the MBE assembly on rank 0.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Honest inputs: every evaluation (warm-up and timed) sees the cluster under a fresh seeded rigid motion (rotation +
translation), so all coordinates differ bit for bit from the previous evaluation and nothing keyed on geometry (ERI
list / block-sharing plan, fragment layouts) can be reused; the energy is invariant under the motion, which the line
reports as `rigid_motion_energy_spread`.  `cold_ms` is the very first evaluation of the process (term-list generation,
topology build, pool allocation, kernel loading included): the one-shot cost of an MBE-2 energy.

Multi-GPU: fragments are independent units -- no collective sits on the data path.  Default (`--scaling weak`): per-GPU work
is fixed, every rank evaluates the 2080 SCFs of its own copy of the cluster (its own rigid motion; all copies have the
same MBE-2 energy, which the line checks) and `value` is the iterations of all ranks over the slowest rank's time.
`--scaling strong`: the ONE cluster's cost-sorted term list round-robin over the ranks, one all-reduce of the zero-padded
energy vector -- total work fixed; at 2080 fragments that regime is latency-bound.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- the dominant kernel of THIS run (J/K stream, integral stage, XC quadrature or SCF step, whichever
                  took the most HIP-event time inside the timed region) against its bound, plus `stages` with the
                  figure of every stage
  cpu_baseline -- the oracle (a CPU port of the reference's libcint path) on this box's host cores the way the
                  reference runs an MBE job: one single-threaded process per core, fragments handed out one at a
                  time (mqc_many_body_expansion.f90:455-461), over a bounded sample of the same fragments; the
                  sample's energies are compared with the GPU's (`parity_max_abs_diff`).
"""
import argparse
import json
import os
import sys
import time

# torch initialises the HIP runtime before the engine's library is loaded: the hardware-queue request the library makes
# for itself (engine.cpp: ask_for_hardware_queues; two lanes of four to eight streams) has to be in the environment here
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP64_PEAK_TFLOPS = 78.6        # dense FP64 (vector = matrix) peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--side", type=int, default=4, help="water lattice side (4 -> 64 waters)")
    ap.add_argument("--basis", default="cc-pvdz")
    ap.add_argument("--functional", default="", help="empty = RHF (configs[2]); e.g. b3lyp")
    ap.add_argument("--df", action="store_true", help="density-fitted J/K with the repo's even-tempered auxiliary set")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo for CPU rehearsal)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every rank evaluates the MBE-2 of its own copy of the cluster (per-GPU work fixed); "
                         "strong = the ONE cluster's term list round-robin over the ranks (total work fixed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dimers-per-core", type=int, default=2, help="CPU-baseline sample: dimers per host core")
    ap.add_argument("--cpu-procs", type=int, default=0, help="CPU-baseline worker processes (0 = the cores this process may use, at most 32)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra B3LYP / DF / single-call measurements")
    ap.add_argument("--fixed-geometry", action="store_true", help="re-evaluate bit-identical coordinates (round-1 behaviour: caches hit)")
    ap.add_argument("--energy-tol", type=float, default=1e-10)
    ap.add_argument("--density-tol", type=float, default=1e-8)
    ap.add_argument("--schwarz-tol", type=float, default=1e-12,
                    help="Schwarz threshold of the in-core ERI build (0 = every quartet); 1e-12 moves a fragment energy by < 1e-11 Eh")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(job):
    """One fragment through the oracle in a single-threaded worker process (rank-per-fragment)."""
    basis, z, coords, nelec, e_tol, d_tol = job[:6]
    functional = job[6] if len(job) > 6 else ""
    from metalquicha_amd.methods import PhysicalFragment
    from oracle import scf_oracle as so
    from tests.helpers import oracle_mol
    frag = PhysicalFragment(np.asarray(z), np.asarray(coords))
    mol = oracle_mol(basis, frag)
    xc = None
    if functional:
        from oracle import xc_oracle
        xc = xc_oracle.XCOracle(mol, functional, 3)
    r = so.run_rhf(mol, int(nelec), 100, e_tol, d_tol, xc=xc)
    return float(r.energy), int(r.iterations)


def cpu_baseline(system, terms, basis, gpu_energies, args, functional="", dimers_per_core=None):
    """P single-threaded worker processes pull fragments from a queue, as the reference's MBE ranks do."""
    import multiprocessing as mp
    from metalquicha_amd import mbe
    from oracle import scf_oracle as so
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    procs = args.cpu_procs if args.cpu_procs > 0 else max(1, min(avail, 32))
    per_core = args.cpu_dimers_per_core if dimers_per_core is None else dimers_per_core
    dimers = [i for i, t in enumerate(terms) if len(t) == 2][: procs * per_core]
    monos = [i for i, t in enumerate(terms) if len(t) == 1][: max(2, procs // 4)]
    sample = dimers + monos
    jobs = []
    for i in sample:
        frag = mbe.build_fragment(system, terms[i])
        jobs.append((basis, frag.element_numbers.tolist(), frag.coordinates.tolist(), int(frag.nelec), args.energy_tol, args.density_tol, functional))
    so.build_oracle_lib()
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"                       # inherited by the spawned workers: one thread each
    try:
        ctx = mp.get_context("spawn")
        with ctx.Pool(procs) as pool:
            pool.map(_cpu_worker, jobs[-1:] * procs, chunksize=1)      # start the workers: numpy and the C library load on a monomer
            t0 = time.perf_counter()
            out = pool.map(_cpu_worker, jobs, chunksize=1)
            dt = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    e_cpu = np.array([o[0] for o in out]); iters = int(sum(o[1] for o in out))
    diff = float(np.max(np.abs(e_cpu - gpu_energies[sample]))) if gpu_energies is not None else None
    return {"value": iters / dt, "unit": "SCF iterations/s", "cores": procs, "kind": "port",
            "sample": "%d water dimers + %d monomers of the same cluster, %s/%s, oracle (C McMurchie-Davidson integrals, "
                      "numpy linear algebra%s), %d single-threaded worker processes pulling one fragment at a time "
                      "(the reference's rank-per-fragment scheme), %.1f s" % (len(dimers), len(monos), (functional.upper() or "RHF"), basis,
                                                                              ", level-3 grid quadrature in 4096-point blocks" if functional else "", procs, dt),
            "seconds": dt, "fragments": len(sample), "host_cores_visible": avail,
            "parity_max_abs_diff": diff, "parity_fragments": len(sample)}


def committed_pmc():
    """Figures from the committed rocprofv3 --pmc passes of THIS command (profiles/r03_pmc_summary.json, written by
    scripts/pmc_summary.py from separate counter runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
    passes of their own): HBM bytes per launch of the J/K kernel, FP64 flop of the integral stage, MFMA-busy fractions.
    None when absent."""
    for name in ("r03_pmc_summary.json", "r02_pmc_summary.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
                d["_source"] = "profiles/" + name
                return d
        except Exception:
            continue
    return None


def rigid_motion(system, step, seed=977):
    """The cluster under a seeded rotation + translation: same molecule, every coordinate bit different."""
    from metalquicha_amd import mbe
    rng = np.random.default_rng(seed + step)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    R = np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)],
                  [2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b)],
                  [2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d]])
    shift = rng.uniform(-4.0, 4.0, size=3)
    coords = R @ system.coordinates + shift[:, None]
    return mbe.FragmentedSystem(system.element_numbers, coords, system.monomers, system.charges, system.multiplicities)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MQC_BENCH_DEVICE"):          # rehearsal on a one-GPU box: all ranks on one device (with --backend gloo)
        local_rank = int(os.environ["MQC_BENCH_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    use_cuda_tensors = False
    if world > 1:
        import torch
        import torch.distributed as dist
        use_cuda_tensors = args.backend == "nccl"
        if use_cuda_tensors:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=args.backend, init_method="env://", rank=rank, world_size=world)

    from metalquicha_amd import capi, mbe, methods

    capi.get_context(local_rank)
    system0 = mbe.water_cluster(args.side)
    settings = methods.ScfSettings(basis_set=args.basis, functional=args.functional, guess="gwh", energy_tol=args.energy_tol,
                                   density_tol=args.density_tol, device_rank=local_rank, density_fitting=args.df,
                                   aux_basis_set="mqc-even-tempered-jkfit", schwarz_tol=args.schwarz_tol)

    def barrier():
        if world > 1:
            dist.barrier()
            if use_cuda_tensors:
                torch.cuda.synchronize()

    weak = args.scaling == "weak"
    rank_totals = []          # weak scaling: every rank's MBE-2 energy of every evaluation (one cluster, moved differently)

    def evaluate(system, st, terms):
        if weak:
            # per-GPU work fixed: this rank evaluates ALL terms of its own copy of the cluster -- the units are independent
            # fragments, nothing crosses between the GPUs on the data path
            run = mbe.run_mbe(system, st, level=2, rank=0, world=1, terms=terms)
        else:
            run = mbe.run_mbe(system, st, level=2, rank=rank, world=world, terms=terms)
        if run.errors:
            raise RuntimeError("fragment failed: " + run.errors[0])
        energies, iters = run.energies, run.iterations.astype(np.float64)
        if world > 1 and weak:
            total, by_order, _ = mbe.compute_mbe(terms, energies)
            # the only inter-GPU traffic: every rank's iteration count and MBE-2 energy, gathered
            mine = torch.tensor([float(np.sum(iters)), float(total)], dtype=torch.float64)
            if use_cuda_tensors:
                mine = mine.cuda()
            gathered = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)
            if use_cuda_tensors:
                torch.cuda.synchronize()
            both = np.array([g.cpu().numpy() for g in gathered])
            rank_totals.extend(float(x) for x in both[:, 1])
            return float(both[0, 1]), energies, float(np.sum(both[:, 0]))
        if world > 1:
            # the only inter-GPU traffic: one all-reduce of the zero-padded per-fragment vectors
            buf = torch.from_numpy(np.concatenate([energies, iters]))
            if use_cuda_tensors:
                buf = buf.cuda()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            if use_cuda_tensors:
                torch.cuda.synchronize()
            both = buf.cpu().numpy()
            energies, iters = both[: len(terms)], both[len(terms):]
        total, by_order, _ = mbe.compute_mbe(terms, energies)
        return total, energies, float(np.sum(iters))

    def moved(step):
        if args.fixed_geometry:
            return system0
        return rigid_motion(system0, step + (100003 * rank if (weak and world > 1) else 0))

    # ---- the cold evaluation: first call of the process, unmoved cluster, term list generated inside
    barrier()
    t0 = time.perf_counter()
    terms = mbe.generate_mbe_term_list(system0, 2)
    e_cold, energies0, _ = evaluate(system0, settings, terms)
    barrier()
    cold_s = time.perf_counter() - t0
    for w in range(max(args.warmup - 1, 0)):
        evaluate(moved(1000 + w), settings, terms)
    methods.get_stats()           # reset the engine's counters: only the timed region is measured
    systems = [moved(s) for s in range(args.steps)]       # the inputs of the timed steps exist before the clock starts
    barrier()
    t0 = time.perf_counter()
    tot_iters = 0.0
    e_steps = []
    for s in range(args.steps):
        e_total, _, it = evaluate(systems[s], settings, terms)
        e_steps.append(e_total)
        tot_iters += it
    barrier()
    elapsed = time.perf_counter() - t0
    main_totals = list(rank_totals)       # cold, warm-up and timed evaluations of the headline method on every rank
    st = methods.get_stats()
    if world > 1:
        tmax = torch.tensor([elapsed, cold_s], dtype=torch.float64)
        if use_cuda_tensors:
            tmax = tmax.cuda()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, cold_s = float(tmax[0].item()), float(tmax[1].item())

    # ---- the north star's own workload as a first-class block: the same cluster and MBE-2 term list with B3LYP (XC on
    # the level-3 pruned grid, 20 % exact exchange from the same in-core tensor): one warm-up + two timed evaluations on
    # fresh geometries, its own engine statistics (-> its own roofline) -- never part of `value`
    b3 = None
    if not args.no_secondary and not args.functional and not args.df:
        sb = methods.ScfSettings(basis_set=args.basis, functional="b3lyp", guess="gwh", energy_tol=args.energy_tol,
                                 density_tol=args.density_tol, schwarz_tol=args.schwarz_tol, device_rank=local_rank)
        _, e_b3_frag, _ = evaluate(system0, sb, terms)       # warm-up on the unmoved cluster: its fragment energies are what the CPU sample is compared with
        methods.get_stats()
        b3_systems = [moved(1901 + k) for k in range(2)]
        barrier()
        t1 = time.perf_counter()
        b3_iters = 0.0
        b3_e = []
        for sysb in b3_systems:
            eb, _, itb = evaluate(sysb, sb, terms)
            b3_e.append(eb); b3_iters += itb
        barrier()
        b3_dt = time.perf_counter() - t1
        b3_st = methods.get_stats()
        if world > 1:
            t2 = torch.tensor([b3_dt], dtype=torch.float64)
            if use_cuda_tensors:
                t2 = t2.cuda()
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            b3_dt = float(t2.item())
        b3 = {"dt": b3_dt, "iters": b3_iters, "stats": b3_st, "energies": b3_e, "frag_energies": e_b3_frag, "steps": len(b3_systems)}

    # ---- secondary measurements (never part of `value`): the north star's B3LYP target and the density-fitted path on
    # the same cluster (one warm-up + one timed evaluation each, moved geometry), and the literal drop-in: ONE fragment
    # per mqc_hip_scf_run call, as an unchanged do_fragment_work would drive the engine
    secondary = None
    if not args.no_secondary and not args.functional and not args.df and world == 1:     # single-GPU measurements: the N = 1 line carries them
        secondary = {}
        for label, kw in (("rhf_density_fitted", dict(density_fitting=True)),
                          ("b3lyp_density_fitted", dict(functional="b3lyp", density_fitting=True))):
            s2 = methods.ScfSettings(basis_set=args.basis, guess="gwh", energy_tol=args.energy_tol, density_tol=args.density_tol,
                                     schwarz_tol=args.schwarz_tol, device_rank=local_rank, aux_basis_set="mqc-even-tempered-jkfit", **kw)
            evaluate(moved(2000), s2, terms)
            methods.get_stats()
            sysb = moved(2001)
            barrier()
            t1 = time.perf_counter()
            _, _, its = evaluate(sysb, s2, terms)
            barrier()
            dt = time.perf_counter() - t1
            st2 = methods.get_stats()
            if world > 1:
                t2 = torch.tensor([dt], dtype=torch.float64)
                if use_cuda_tensors:
                    t2 = t2.cuda()
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                dt = float(t2.item())
            secondary[label] = {"mbe2_wall_s": dt, "scf_iterations_per_s": its / dt, "scf_iterations": its,
                                "xc_kernel_seconds": st2.xc_kernel_seconds, "xc_points": st2.xc_points,
                                "xc_algorithmic_tflops": (st2.xc_flops / st2.xc_kernel_seconds / 1e12) if st2.xc_kernel_seconds > 0 else None,
                                "xc_frac_of_fp64_peak": (st2.xc_flops / st2.xc_kernel_seconds / 1e12 / FP64_PEAK_TFLOPS) if st2.xc_kernel_seconds > 0 else None,
                                "jk_kernel_seconds": st2.fock_kernel_seconds,
                                "df_algorithmic_gbs": (st2.df_bytes / st2.fock_kernel_seconds / 1e9) if (kw.get("density_fitting") and st2.fock_kernel_seconds > 0) else None,
                                # the bytes the J/K kernel really reads: the fitted tensor is stored packed [naux][npair], read once
                                "df_stored_tensor_gbs": (st2.fock_bytes / st2.fock_kernel_seconds / 1e9) if (kw.get("density_fitting") and st2.fock_kernel_seconds > 0) else None,
                                "df_stored_tensor_frac_of_hbm_peak": (st2.fock_bytes / st2.fock_kernel_seconds / 1e9 / HBM_PEAK_GBS) if (kw.get("density_fitting") and st2.fock_kernel_seconds > 0) else None,
                                "df_algorithmic_tflops": (st2.df_flops / st2.fock_kernel_seconds / 1e12) if (kw.get("density_fitting") and st2.fock_kernel_seconds > 0) else None,
                                "integral_stage_seconds": st2.eri_kernel_seconds, "scf_step_kernel_seconds": st2.scf_step_seconds}
        if rank == 0:
            # single-call drop-in: the first 48 dimers and 16 monomers, one mqc_hip_scf_run each
            sysc = moved(3000)
            sub = [t for t in terms if len(t) == 2][:48] + [t for t in terms if len(t) == 1][:16]
            frags = [mbe.build_fragment(sysc, t) for t in sub]
            methods.run_hip_scf(settings, frags[0]); methods.run_hip_scf(settings, frags[-1])
            t1 = time.perf_counter()
            its = 0
            for f in frags:
                r = methods.run_hip_scf(settings, f)
                if r.has_error:
                    raise RuntimeError("single-call fragment failed: " + r.error_message)
                its += r.scf_iterations
            dt = time.perf_counter() - t1
            secondary["single_call"] = {"scf_iterations_per_s": its / dt, "fragments": len(frags), "seconds": dt,
                                        "ms_per_fragment": 1e3 * dt / len(frags),
                                        "note": "one fragment per mqc_hip_scf_run (unchanged do_fragment_work): a batch of one"}
            # the embedded callers on the same cluster: FMO2 with the Mulliken point-charge field
            # through fmo.run_fmo2 -- monomer passes to self-consistency + the pair phase, fresh geometry, one warm-up
            try:
                from metalquicha_amd import fmo as _fmo
                fst = methods.ScfSettings(basis_set=args.basis, guess="gwh", energy_tol=1e-9, density_tol=1e-7, device_rank=local_rank)
                _fmo.run_fmo2(moved(3100), fst, expansion="fmo")
                t1 = time.perf_counter()
                fr = _fmo.run_fmo2(moved(3101), fst, expansion="fmo")
                dt = time.perf_counter() - t1
                secondary["fmo2_point_charge_field"] = {
                    "wall_s": dt, "scf_iterations": fr.scf_iterations, "scf_iterations_per_s": fr.scf_iterations / dt,
                    "outer_passes": fr.outer_iterations, "converged": bool(fr.converged), "energy_hartree": fr.energy,
                    "note": "mqc_libcint_fmo.f90 run_fmo2 with esp = ptc, Mulliken charges; every pass one engine batch"}
            except Exception as e:      # a secondary must not take the headline line down with it
                secondary["fmo2_point_charge_field"] = {"error": str(e)}
            # BASELINE.json configs[1]: ONE benzene, B3LYP/cc-pVDZ, density-fitted J/K (n = 114, grid level 3); a fresh
            # rotation per repeat so that nothing is served from a geometry-keyed cache
            import numpy as _np
            rcc, rch = 1.397, 1.084
            sym = ["C"] * 6 + ["H"] * 6
            xyz0 = _np.array([[rcc * _np.cos(_np.pi / 3 * k), rcc * _np.sin(_np.pi / 3 * k), 0.0] for k in range(6)] +
                             [[(rcc + rch) * _np.cos(_np.pi / 3 * k), (rcc + rch) * _np.sin(_np.pi / 3 * k), 0.0] for k in range(6)])
            bst = methods.ScfSettings(basis_set="cc-pvdz", functional="b3lyp", density_fitting=True, aux_basis_set="mqc-even-tempered-jkfit",
                                      energy_tol=args.energy_tol, density_tol=args.density_tol, guess="gwh")
            rngb = _np.random.default_rng(5)
            times, iters, e_b = [], 0, None
            for rep in range(3):
                q, _ = _np.linalg.qr(rngb.normal(size=(3, 3)))
                frag_b = methods.PhysicalFragment.from_angstrom(sym, (xyz0 @ q.T).tolist())
                t1 = time.perf_counter()
                rb = methods.run_hip_scf(bst, frag_b)
                dtb = time.perf_counter() - t1
                if rb.has_error:
                    raise RuntimeError("benzene secondary failed: " + rb.error_message)
                if rep > 0:
                    times.append(dtb); iters += rb.scf_iterations
                e_b = rb.energy.scf
            secondary["benzene_b3lyp_df_single_fragment"] = {
                "seconds_per_scf": sum(times) / len(times), "scf_iterations_per_s": iters / sum(times), "scf_iterations": iters // len(times),
                "energy_hartree": e_b, "n_ao": 114,
                "note": "BASELINE configs[1]; auxiliary set mqc-even-tempered-jkfit; one fragment = a batch of one (latency-bound stages)"}

    if rank == 0:
        n_steps = max(args.steps, 1)
        pmc = committed_pmc() if (args.side == 4 and args.basis == "cc-pvdz" and not args.df and not args.functional and world == 1) else None
        # J/K stream over the dimer batch (launches that move >= 1 GiB), HIP events on the engine's own stream
        big_s, big_b, big_n = st.fock_big_seconds, st.fock_big_bytes, int(st.fock_big_launches)
        if big_n == 0:       # small workloads (--side 2): all launches
            big_s, big_b, big_n = st.fock_kernel_seconds, st.fock_bytes, int(st.fock_launches)
        jk_name = "df_jk_mfma_kernel" if args.df else "jk_incore_kernel"
        jk_layout = None
        if not args.df and big_n > 0:
            # which tensor the dimer launches streamed: the square (npair^2 doubles per fragment and launch) or its lower
            # triangle (kern_fock.hip jk_tri_kernel, npair (npair + 1) / 2) -- told from the bytes the engine counted
            from metalquicha_amd.basis import build_flat_basis
            n_d = 2 * int(build_flat_basis(args.basis, [8, 1, 1]).nao)
            npd = n_d * (n_d + 1) // 2
            n_dimers = len(terms) - system0.n_monomers
            per_frag = big_b / big_n / max(n_dimers if weak else n_dimers // world, 1) / 8.0
            if per_frag < 0.75 * npd * npd:
                jk_name = "jk_tri_kernel"
                jk_layout = ("lower triangle of the pair matrix in blocks of row pairs (zero-padded to whole shell rows); the kernel skips pair rows "
                             "the Schwarz bounds prove zero and counts the 1 KiB chunks it reads: %d doubles per fragment and launch on average "
                             "(stored: about 729 k with the padding; the bare triangle npair (npair + 1) / 2 = %d, the square %d)"
                             % (int(round(per_frag)), npd * (npd + 1) // 2, npd * npd))
            else:
                jk_layout = "square pair matrix, npair^2 = %d doubles per fragment" % (npd * npd)
        jk_bytes = st.fock_bytes if args.df else big_b      # --df: the packed fitted tensor the kernel really reads (8 npair A per fragment-iteration)
        jk_secs = st.fock_kernel_seconds if args.df else big_s
        jk_launches = int(st.fock_launches) if args.df else big_n
        stages = {
            "jk": {"kernel": jk_name, "bound": "hbm", "seconds": st.fock_kernel_seconds, "big_launch_seconds": jk_secs,
                   "launches": jk_launches, "algorithmic_bytes": jk_bytes, "tensor_layout": jk_layout,
                   "achieved_gbs": (jk_bytes / jk_secs / 1e9) if jk_secs > 0 else None,
                   "frac_of_hbm_peak": (jk_bytes / jk_secs / 1e9 / HBM_PEAK_GBS) if jk_secs > 0 else None},
            "eri": {"kernel": "eri class kernels (integral stage, all streams)", "bound": "fp64 valu", "seconds": st.eri_kernel_seconds,
                    "quartets_formed": int(st.eri_survivors), "canonical_quartets": int(st.eri_quartets),
                    "quartets_per_s": (st.eri_survivors / st.eri_kernel_seconds) if st.eri_kernel_seconds > 0 else None,
                    # FP64 flop of the eri_* / schwarz_* kernels from the SQ instruction counters of a separate --pmc pass of
                    # this command (profiles/r02_pmc_summary.json), divided by THIS run's stage time per evaluation
                    "fp64_tflops_from_counters": ((pmc["eri_fp64_flop_total"] / pmc.get("eri_fp64_evaluations_in_pass", 1)
                                                   / (st.eri_kernel_seconds / n_steps) / 1e12)
                                                  if (pmc and pmc.get("eri_fp64_flop_total") and st.eri_kernel_seconds > 0) else None),
                    "hbm_bytes_per_evaluation_from_counters": ((pmc["eri_stage_hbm_bytes_total"] / pmc.get("eri_fp64_evaluations_in_pass", 1))
                                                               if (pmc and pmc.get("eri_stage_hbm_bytes_total")) else None)},
            "xc": {"kernel": "xc quadrature", "bound": "mfma", "seconds": st.xc_kernel_seconds, "points": st.xc_points,
                   "algorithmic_tflops": (st.xc_flops / st.xc_kernel_seconds / 1e12) if st.xc_kernel_seconds > 0 else None,
                   "frac_of_fp64_peak": (st.xc_flops / st.xc_kernel_seconds / 1e12 / FP64_PEAK_TFLOPS) if st.xc_kernel_seconds > 0 else None},
            "scf_step": {"kernel": "scf_step_kernel", "bound": "latency", "seconds": st.scf_step_seconds},
        }
        dominant = max(("jk", "eri", "xc", "scf_step"), key=lambda k: stages[k]["seconds"] or 0.0)
        if dominant == "xc":
            ach = stages["xc"]["algorithmic_tflops"]
            roof = {"bound": "mfma", "kernel": "xc_mfma_kernel", "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": (ach / FP64_PEAK_TFLOPS) if ach else None, "traffic": None,
                    "kernel_seconds": st.xc_kernel_seconds, "algorithmic_flops": st.xc_flops}
        else:
            # the integral stage has no byte/flop unit of its own (quartets/s); when it or the SCF step leads, the
            # roofline block still prices the largest kernel that HAS a bound -- the J/K stream -- and says which stage led
            ach = stages["jk"]["achieved_gbs"]
            roof = {"bound": "hbm", "kernel": jk_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (ach / HBM_PEAK_GBS) if ach else None,
                    # counters of the committed --pmc passes, only when they belong to the kernel that ran here
                    "traffic": ((pmc or {}).get("jk_hbm_bytes_per_launch")
                                if str((pmc or {}).get("jk_kernel", "jk_incore_kernel")).startswith(jk_name) else None),
                    "kernel_seconds": jk_secs, "launches": jk_launches, "algorithmic_bytes": jk_bytes,
                    "algorithmic_bytes_per_launch": (jk_bytes / jk_launches) if jk_launches else None,
                    "avg_launch_ms": (1e3 * jk_secs / jk_launches) if jk_launches else None}
        roof["dominant_stage"] = dominant
        roof["stages"] = stages
        spread = float(np.max(np.abs(np.array(e_steps + main_totals + [e_cold]) - e_cold))) if e_steps else None
        jobs = world if weak else 1
        line = {
            "metric": "SCF iterations/s (whole job); MBE-2 wall time @64 fragments",
            "value": tot_iters / elapsed,
            "unit": "SCF iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / n_steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "(H2O)%d MBE-2 %s/%s, %s, %d SCFs (%d monomers + %d dimers), GWH guess, e_tol %.0e d_tol %.0e%s; %s"
                                   % (system0.n_monomers, (args.functional.upper() or "RHF"), args.basis,
                                      "density-fitted J/K (even-tempered aux)" if args.df else
                                      ("exact in-core ERIs, Schwarz-screened at %.0e" % args.schwarz_tol if args.schwarz_tol > 0 else "exact in-core ERIs, unscreened"),
                                      len(terms), system0.n_monomers, len(terms) - system0.n_monomers, args.energy_tol, args.density_tol,
                                      ", grid level 3 (pruned)" if args.functional else "",
                                      "bit-identical geometry every step" if args.fixed_geometry else "fresh rigid motion of the cluster every step"),
                       "fragments": len(terms) * jobs,
                       "parallelism": ("one MBE-2 job of %d independent SCFs per GPU, %d GPU(s): every rank its own rigid motion of the cluster, "
                                       "no data-path collective (one all-gather of two numbers per evaluation)" % (len(terms), world)) if weak
                                      else "fragments round-robin over %d GPU(s)" % world},
            "mbe2_wall_s": elapsed / n_steps,
            "cold_ms": 1e3 * cold_s,
            "mbe2_energy_hartree": e_cold,
            "rigid_motion_energy_spread": spread,
            "scf_iterations_per_step": tot_iters / n_steps,
            "engine_seconds": {"setup": st.t_setup, "int1e+orthogonaliser": st.t_int1e, "eri": st.t_eri,
                               "scf_loop": st.t_fock, "fetch": st.t_scf_step, "total": st.t_total},
            "roofline": roof,
        }
        if b3 is not None:
            bs = b3["stats"]
            # XC stage: the engine's HIP events around its quadrature launches (density, functional and potential
            # kernels of the split quadrature).  Two flop counts: SURVEY 8d's algorithmic 8 P n^2, and the MFMA flop the
            # kernels actually issue, 4 P n^2 (X = D chi and A += a chi^T; the GGA contraction uses ONE combined vector a)
            xc_s = bs.xc_kernel_seconds
            mfma_tf = (0.5 * bs.xc_flops / xc_s / 1e12) if xc_s > 0 else None
            pm = committed_pmc() or {}
            mf = None
            for kname, v in pm.get("passes", {}).get("sq_b3lyp", {}).get("kernels", {}).items():
                if kname.startswith("xc_density_kernel") and "mfma_busy_fraction" in v:
                    mf = {"kernel": kname, "mfma_busy_fraction": v["mfma_busy_fraction"], "mfma_tflops": v.get("mfma_tflops"), "source": pm.get("_source")}
            line["b3lyp"] = {
                "workload": "(H2O)%d MBE-2 B3LYP/%s (libxc hyb_gga_xc_b3lyp: 0.08 Slater + 0.72 B88 + 0.19 VWN-RPA + 0.81 LYP, 20 %% exact exchange), "
                            "level-3 pruned grid, exact in-core ERIs Schwarz-screened at %.0e, %d SCFs, GWH guess, e_tol %.0e d_tol %.0e; fresh rigid motion per evaluation"
                            % (system0.n_monomers, args.basis, args.schwarz_tol, len(terms), args.energy_tol, args.density_tol),
                "value": b3["iters"] / b3["dt"], "unit": "SCF iterations/s", "steps": b3["steps"],
                "mbe2_wall_s": b3["dt"] / b3["steps"], "scf_iterations_per_step": b3["iters"] / b3["steps"],
                "mbe2_energy_hartree": b3["energies"][-1],
                "stage_seconds_per_step": {"xc": xc_s / b3["steps"], "jk": bs.fock_kernel_seconds / b3["steps"],
                                           "eri": bs.eri_kernel_seconds / b3["steps"], "scf_step": bs.scf_step_seconds / b3["steps"]},
                "roofline": {"bound": "mfma", "kernel": "xc_density_kernel + xc_functional_kernel + xc_potential_kernel (XC stage)",
                             "achieved": mfma_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": (mfma_tf / FP64_PEAK_TFLOPS) if mfma_tf else None, "traffic": None,
                             "flops_counted": "MFMA flop issued: 4 P n^2 per launch (P = grid points x active fragments)",
                             "algorithmic_tflops_8Pn2": (bs.xc_flops / xc_s / 1e12) if xc_s > 0 else None,
                             "kernel_seconds": xc_s, "points": bs.xc_points, "mfma_counters": mf},
            }
            if not args.no_cpu_baseline and world == 1:
                cb = cpu_baseline(system0, terms, args.basis, b3["frag_energies"], args, functional="b3lyp", dimers_per_core=1)
                line["b3lyp"]["cpu_baseline"] = cb
                line["b3lyp"]["parity_max_abs_diff"] = cb["parity_max_abs_diff"]
        line["secondary"] = secondary
        pm_all = committed_pmc()
        if pm_all and "passes" in pm_all:
            # MFMA-utilisation counters (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F64) of the B3LYP and DF commands,
            # collected in their own rocprofv3 --pmc passes and committed under profiles/
            def _mf(tag, sub):
                for k, v in pm_all["passes"].get(tag, {}).get("kernels", {}).items():
                    if k.startswith(sub) and "mfma_busy_fraction" in v:
                        return {"kernel": k, "mfma_busy_fraction": v["mfma_busy_fraction"], "mfma_tflops": v["mfma_tflops"],
                                "valu_fp64_tflops": v.get("valu_fp64_tflops"), "peak_tflops": FP64_PEAK_TFLOPS}
                return None
            line["mfma_counters"] = {"xc": _mf("sq_b3lyp", "xc_tile_kernel"), "df_jk": _mf("sq_df", "df_jk_mfma_kernel"),
                                     "df_fit": _mf("sq_df", "df_fit_mfma_kernel"), "source": pm_all.get("_source")}
        if not args.no_cpu_baseline and world == 1 and not args.functional and not args.df:
            line["cpu_baseline"] = cpu_baseline(system0, terms, args.basis, energies0, args)
            line["parity_max_abs_diff"] = line["cpu_baseline"]["parity_max_abs_diff"]
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    capi.finalize()


if __name__ == "__main__":
    main()
