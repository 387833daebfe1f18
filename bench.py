#!/usr/bin/env python3
"""bench.py — env-steps/sec of the hot path: task=go2 on flat terrain, 4096 envs per MI355X (BASELINE.json configs[1]).

One "step" = one full PPO iteration exactly as OnPolicyRunner.learn runs it (on_policy_runner.py:113-172):
24 rollout steps x 4096 envs (policy inference + the fused HIP env step + storage), GAE + advantage normalisation,
5 epochs x 4 mini-batches of PPO (forward, backward, grad clip, Adam).  Nothing is skipped, fp32 throughout.
value = num_gpus * 4096 * 24 * K / elapsed  — the reference's own "Computation: N steps/s" definition (:194).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Extra keys: "roofline" (the fused env-step kernel against HBM), "cpu_baseline" (the
CPU oracle + torch-CPU PPO on the node's host cores, bounded sample), "collection_only" (24*N/collection_time).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_FLAT = 2936          # algorithmic bytes per env-step on a plane: 778 read + 2158 written
ALGO_BYTES_ROUGH = 2936 + 187 * 3 * 2 + 13 * 4 * 4               # SURVEY 8d ALGO_BYTES_HF = 4266: + 3 int16 samples per scan point, 13 contact shapes x 4 substeps x 4 B
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec (MI355X_MICROARCH.md)
NUM_ENVS = 4096


def parse():
    p = argparse.ArgumentParser(
        description="env-steps/s of full PPO iterations (24 rollout steps + GAE + 5 x 4 mini-batch updates) on the HIP env library; prints ONE JSON line.",
        epilog="GPU only: there is no --sim_device cpu path (the CPU restatement under oracle/ is test infrastructure).  Fields beyond the driver's "
               "contract: collection_only (rollout without the update), graphs (whether rollout / update ran as replayed HIP graphs; the run fails "
               "if capture degraded), rccl_ranks + collectives_per_iteration (process-group size actually used and all-reduces counted in the timed "
               "region), roofline (HBM view of go2_step_kernel: algorithmic bytes / live HIP-event kernel time; traffic + valu_issue only when "
               "profiles/ holds counters for the loaded library's sha256), cpu_baseline (oracle env + torch-CPU PPO at num_envs 64 and 4096 on this "
               "host; a reported baseline, rank 0 at N=1 only).")
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=100)     # 100 iterations = 9.8 M env-steps, ~3 s on one MI355X
    p.add_argument("--warmup", type=int, default=20)     # resets / pushes / resamples reach their steady-state rates
    p.add_argument("--num-envs", type=int, default=NUM_ENVS, help="envs PER GPU (weak scaling)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--task", default="go2_flat", choices=["go2_flat", "go2", "go2_flat_cts", "go2_flat_moe_cts", "go2_cts", "go2_moe_cts", "go2_moe_ng_cts", "go2_ac_moe_cts", "go2_dual_moe_cts", "go2_mcp_cts"],
                   help="go2_flat is the BASELINE workload; the others are extra lines (rough curriculum terrain, CTS / MoE-CTS algorithms), never the headline")
    return p.parse_args()


CPU_THREADS = 16          # default OpenMP / torch thread count outside the sweep (N = 64: more only oversubscribes 64 envs)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; os.cpu_count() when that cannot be read"""
    try:
        seen, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                seen.add((phys, line.split(":", 1)[1].strip()))
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def _set_cpu_threads(n):
    """OpenMP threads of the oracle (libgomp is already loaded: omp_set_num_threads, not the environment) and torch's intra-op threads"""
    import torch
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass
    torch.set_num_threads(int(n))


def cpu_baseline(task="go2_flat", sizes=(64, NUM_ENVS), iters=3):
    """The same PPO iteration on the host CPU: the plain-C oracle (OpenMP over envs) as the env + torch-CPU PPO, at
    N = 64 (BASELINE config 1) and N = 4096 (the headline size), collection-only and total.  The reference's own --sim_device=cpu path
    cannot run anywhere (Isaac Gym is absent), so this is the build's CPU restatement of the same path ("kind": "port").
    At the headline size the thread count is SWEPT upwards (8 / 16 / 32 / 64 / physical cores, stopping where it stops paying): one warm-up + one
    timed iteration each, then `iters` timed iterations at the best — the reported value is the host's best, with its thread count next to nproc.
    Test-infrastructure code, timed only here, never inside the GPU region."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_oracle
    from go2_rl_gym_amd.envs import task_registry  # noqa: F401
    from go2_rl_gym_amd.utils import get_args
    nproc, phys = os.cpu_count() or 1, _physical_cores()
    per_size, sweep = {}, {}
    best_threads = min(nproc, CPU_THREADS)
    for n in sizes:
        args = get_args(["--task", task, "--num_envs", str(n), "--sim_device", "cpu", "--rl_device", "cpu", "--headless"])
        env, _ = task_registry.make_env(task, args, lib=load_oracle())
        runner, _ = task_registry.make_alg_runner(env, task, args, log_root=None)
        threads = min(nproc, CPU_THREADS)
        if n >= 1024:
            # ascending thread counts; the sweep stops at a count whose iteration takes more
            # than 12 s (a container may report more hardware threads than its CPU quota gives it: oversubscribed OpenMP pools crawl), and
            # in any case after 60 s — bench.py has to finish within minutes
            t_sweep = time.perf_counter()
            for th in sorted({min(nproc, c) for c in (8, 16, 32, 64, phys)}):
                _set_cpu_threads(th)
                runner.learn(1, init_at_random_ep_len=(not sweep))
                t0 = time.perf_counter()
                runner.learn(1)
                dt = time.perf_counter() - t0
                sweep[str(th)] = {"env_steps_per_s": n * 24 / dt, "collection_only_env_steps_per_s": n * 24 / runner.last_collection_time}
                best = max(v["env_steps_per_s"] for v in sweep.values())
                # (a count that is slower than the best so far ends the sweep only beyond 64 threads: 8 .. 64 are always timed)
                if (th >= 64 and sweep[str(th)]["env_steps_per_s"] < 0.9 * best) or dt > 12.0 or time.perf_counter() - t_sweep > 60.0:
                    break
            threads = best_threads = int(max(sweep, key=lambda k: sweep[k]["env_steps_per_s"]))
            _set_cpu_threads(threads)
        else:
            _set_cpu_threads(threads)
            runner.learn(1, init_at_random_ep_len=True)
        tot = col = 0.0
        for _ in range(iters):
            t0 = time.perf_counter()
            runner.learn(1)
            tot += time.perf_counter() - t0
            col += runner.last_collection_time
        env.close()
        per_size[str(n)] = {"env_steps_per_s": n * 24 * iters / tot, "collection_only_env_steps_per_s": n * 24 * iters / col, "iterations": iters, "seconds": tot, "threads": threads}
    head = per_size[str(sizes[-1])]
    return {"value": head["env_steps_per_s"], "unit": "env-steps/s", "cores": best_threads, "kind": "port", "nproc": nproc, "physical_cores": phys, "omp_threads": best_threads,
            "torch_threads": best_threads, "cpu_model": _cpu_model(), "collection_only": head["collection_only_env_steps_per_s"], "per_num_envs": per_size,
            "thread_sweep_at_%d" % sizes[-1]: sweep,
            "sample": "oracle env (plain C, OpenMP) + torch-CPU PPO; full PPO iterations (24 steps + GAE + 5x4 mini-batches) at num_envs = %s; at num_envs=%d the "
                      "OpenMP/torch thread count is swept (1 warm-up + 1 timed iteration each) and value = %d timed iterations at the best count (`cores`); "
                      "num_envs=64 runs with %d threads" % (" and ".join(str(n) for n in sizes), sizes[-1], iters, min(nproc, CPU_THREADS))}


_STDOUT = sys.stdout


def main():
    a = parse()
    sys.stdout = sys.stderr          # stdout carries ONE line, the JSON: what the mirrored helpers print on the way ("Setting seed: 1", as the reference's set_seed does) goes to stderr
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, CPU_THREADS)))   # read by libgomp (the oracle) at load
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                               # dmabuf IPC for RCCL across processes (host driver requirement)
    os.environ.setdefault("GO2_STRICT_GRAPHS", "1")          # a failed HIP-graph capture raises: no number from a silently degraded (eager) run
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly (`python bench.py --gpus N`): become the launcher — one rank per GPU under torch.distributed.run, as the driver's own command line does;
        # rank 0's JSON line is this process's stdout
        import socket
        import subprocess
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, stdout=_STDOUT, stderr=sys.stderr).returncode)
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d under a launcher with WORLD_SIZE=%d: start it with --nproc-per-node %d (or plainly: bench.py launches its own ranks)" % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the product has no CPU path)")
    backend = os.environ.get("GO2_DIST_BACKEND", "nccl")
    if backend != "nccl":
        # Rehearsal of the N-rank code path on FEWER GPUs than ranks (tests/test_gpu_parity.py: 2 ranks on the one GPU of a test box; RCCL
        # refuses two ranks on one device, gloo stages the collectives through the host).  Same shards, same captured halves, same collective
        # sequence; its throughput means nothing and the line says so ("backend").
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    if world > 1 or os.environ.get("GO2_FORCE_COLLECTIVES", "0") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))      # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from go2_rl_gym_amd.envs import task_registry  # noqa: F401
    from go2_rl_gym_amd.utils import get_args
    N = a.num_envs
    args = get_args(["--task", a.task, "--num_envs", str(N), "--sim_device", dev, "--rl_device", dev, "--headless", "--seed", "1"])
    env, env_cfg = task_registry.make_env(a.task, args, env_offset=rank * N, num_envs_global=world * N)
    torch.manual_seed(1 + rank)       # policy init is broadcast from rank 0; sampling noise differs per shard
    runner, train_cfg = task_registry.make_alg_runner(env, a.task, args, log_root=None)
    env.common_step_counter = 0
    env.update_reward_curriculum(force_update=True)

    runner.learn(a.warmup, init_at_random_ep_len=True)        # untimed: also brings resets/pushes/resamples to steady state
    extra = 0
    while (not all(runner.graphs_captured().values()) or a.warmup + extra < 5) and extra < 8:     # a --warmup shorter than the capture schedule (rollout graph:
        runner.learn(1); extra += 1                                      # 3rd iteration; compute_returns graph: 4th; permutation graph: 3rd): finish capturing, still untimed

    # every RCCL collective issued from Python inside the timed region is counted (the driver can check that RCCL saw N ranks and how often)
    ncoll = {"all_reduce": 0}
    if dist.is_initialized():
        _ar = dist.all_reduce

        def counted_all_reduce(*args_, **kw):
            ncoll["all_reduce"] += 1
            return _ar(*args_, **kw)
        dist.all_reduce = counted_all_reduce
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    col = 0.0
    for _ in range(a.steps):
        runner.learn(1)
        col += runner.last_collection_time
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        dist.all_reduce = _ar
    graphs = runner.graphs_captured()
    if not (graphs["rollout"] and graphs["update"]):
        raise SystemExit("bench.py: HIP-graph capture degraded to eager execution (%r): refusing to report a number for a different execution mode" % (graphs,))
    # The rollout is replayed from a HIP graph, where per-launch events cannot be recorded; so the dominant kernel is timed
    # right after the timed region, live, with HIP events on the stream it is launched on: 48 eager env steps from the same
    # (steady-state) simulator state under the current policy.  profiles/*kernel_stats.csv of the same command must agree.
    env.lib.go2sim_enable_timing(env.handle, 1)
    with torch.inference_mode():
        for _ in range(48):
            ac = runner.alg.actor_critic
            env.step(ac.act(env.get_observations()) if a.task in ("go2_flat", "go2") else ac.act_inference(env.get_observations()))
    ms, n = C.c_double(), C.c_int64()
    env.lib.go2sim_kernel_time(env.handle, C.byref(ms), C.byref(n))
    env.lib.go2sim_enable_timing(env.handle, 0)
    t = torch.tensor([elapsed, col], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, col = t.tolist()

    if rank == 0:
        total_steps = world * N * 24 * a.steps
        k_ms = ms.value / max(n.value, 1)
        algo = ALGO_BYTES_FLAT if "flat" in a.task else ALGO_BYTES_ROUGH
        achieved = algo * N / (k_ms * 1e-3) / 1e9
        # HBM traffic per launch from the rocprofv3 PMC passes (tools/pmc_pass.sh -> profiles/*_pmc_step_kernel.json; bench.py cannot collect
        # counters on itself).  Quoted ONLY when the profile was taken on exactly the library loaded now (sha256 of the .so), at this size
        # and task; the counters are corrected by the calibration probe measured in the same pass (known bytes, same access pattern).
        import glob
        import hashlib
        from go2_rl_gym_amd import _lib
        lib_sha = hashlib.sha256(open(_lib.HIP_LIB, "rb").read()).hexdigest()[:16]
        traffic, traffic_src, traffic_note = None, None, "no PMC profile of this exact library (sha256 %s) in profiles/" % lib_sha
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_step_kernel.json")))[::-1]:
            pm = json.load(open(f))
            if pm.get("num_envs") == N and pm.get("task", "go2_flat") == a.task and pm.get("lib_sha256_16") == lib_sha:
                traffic, traffic_src = pm["hbm_bytes_per_launch_corrected"], os.path.relpath(f, ROOT)
                traffic_note = "FETCH_SIZE %.0f KB / WRITE_SIZE %.0f KB per launch, corrected by the calibration probe's reported/known = %.2f / %.2f" % (
                    pm["FETCH_SIZE"]["mean_kb"], pm["WRITE_SIZE"]["mean_kb"], pm["calibration_FETCH_SIZE"]["reported_over_known"], pm["calibration_WRITE_SIZE"]["reported_over_known"])
                break
        # VALU-issue view of the same kernel (what actually binds it): lane-instructions per env-step from the SQ PMC pass
        # (tools/sq_pass.sh -> profiles/*_sq_step_kernel.json) x envs / measured kernel time, against 1024 SIMDs x 16 lanes x 2.4 GHz
        valu = None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_step_kernel.json")))[::-1]:
            sq = json.load(open(f))
            if sq.get("lib_sha256_16") != lib_sha or sq.get("num_envs") != N or sq.get("task", "go2_flat") != a.task:
                continue
            per_env = sq["SQ_INSTS_VALU"] * 64 / sq["num_envs"]
            valu = {"lane_instr_per_env_step": per_env, "achieved_Tlane_ops": per_env * N / (k_ms * 1e-3) / 1e12, "peak_Tlane_ops": 1024 * 16 * 2.4e9 / 1e12,
                    "frac": per_env * N / (k_ms * 1e-3) / (1024 * 16 * 2.4e9), "wave_cycles_waiting_frac": sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1),
                    "source": os.path.relpath(f, ROOT)}
            break
        # registers / scratch / LDS as the loaded library's CODE OBJECT states them (llvm-readelf --notes of the unbundled gfx950 object)
        from go2_rl_gym_amd import build as _build
        code_object = _build.kernel_resources(_lib.HIP_LIB)
        out = {
            "metric": "env-steps/sec at 4096 envs (go2 flat); 1/2/4/8-GPU scaling", "value": total_steps / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # what "f32" means in the matrix products (the env step and every element-wise kernel are plain fp32): VERDICT r4 item 4
            "arith": ("learner GEMMs and rollout policy kernel: fp32 operands split exactly into 3 bf16 planes, 6 bf16-MFMA terms (v_mfma_f32_32x32x16_bf16), fp32 accumulate — "
                      "no operand bit dropped; env step: fp32 VALU" if os.environ.get("GO2_GEMM_SPLIT", "1") == "1" else
                      "learner GEMMs and rollout policy kernel: fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate; env step: fp32 VALU"),
            "config": {"workload": "task=%s, num_envs=%d per GPU, full PPO iteration = 24 rollout steps + GAE + 5 epochs x 4 mini-batches"
                                   % ("go2 flat terrain (go2_flat)" if a.task == "go2_flat" else a.task + " (NOT the BASELINE workload)", N),
                       "num_envs_per_gpu": N, "num_steps_per_env": 24, "parallelism": "env-sharded dp%d" % world},
            "collection_only": world * N * 24 * a.steps / col,
            # which kernel evaluated the rollout's networks (go2nn_mlp_arith: 3 = split-operand planes, 1 = fp32 MFMA; null = the torch chains ran): a silent fall-back shows here
            "policy_kernel_arith": (lambda pk: None if pk in (None, False) else {k: getattr(pk, k).arith for k in ("enc_t", "enc_s", "actor", "critic") if getattr(pk, k, None) is not None})(getattr(runner.alg, "_pk", None)),
            "graphs": graphs,      # both halves of every timed iteration were replayed from HIP graphs (bench.py exits non-zero otherwise)
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "backend": (dist.get_backend() if dist.is_initialized() else None),
            "collectives_per_iteration": {"all_reduce": ncoll["all_reduce"] / max(a.steps, 1),
                                          "what": "1 x 24-byte fp64 advantage-statistics all-reduce (rollout_storage.py:137) + 1 flat gradient+KL bucket per mini-batch step (5 x 4)"},
            "roofline": {"bound": "hbm", "kernel": "go2_step_kernel<PHYS|POST>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note, "lib_sha256_16": lib_sha, "algorithmic_bytes_per_launch": algo * N, "kernel_ms": k_ms, "launches": n.value, "algorithmic_bytes_per_env_step": algo,
                         "note": "latency-bound by construction: 4096 envs x 16 lanes = 1024 waves = one wave per SIMD; the binding resource is the dependent-issue "
                                 "latency of the ~16k instructions a wave executes per step, not HBM (DESIGN.md 6)",
                         "code_object": code_object, "valu_issue": valu},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(task=a.task)
            except Exception as e:   # never lose the GPU number to a CPU-side problem
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    else:
        out = None
    env.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    # RCCL prints its version banner through C stdio, which is flushed at exit when stdout is a pipe: flush it now so that the JSON
    # line is the LAST line of the output
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    if out is not None:
        sys.stdout.flush()
        print(json.dumps(out), file=_STDOUT, flush=True)


if __name__ == "__main__":
    main()
