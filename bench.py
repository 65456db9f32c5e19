#!/usr/bin/env python3
"""bench.py -- headline benchmark of the nearest-primitive / sphere-trace path.

Workload (BASELINE.json metric: "Mrays/s ... 1080p 10k-primitive scene"):
config 3 = one 1920x1080 sphere-traced frame with normals over the 10k
primitive synthetic scene (seed 0x5DF00003).  A "step" is one full frame.

  N = 1 : one sdfgpu_render_dev launch per step.
  N > 1 : the frame's 8x8 tiles are dealt round-robin to the ranks (rank r
          renders tiles r, r+N, ...), the tile-compact planes go to rank 0 over
          RCCL (the library's communicator: grouped ncclSend / ncclRecv; if it
          cannot be made, a torch.distributed all-gather) and rank 0 scatters
          them into the frame.  Several frames are in flight on separate streams,
          since 1/N of a frame does not fill a GPU.  The total work per step is
          fixed, so scaling is "strong".

Prints ONE JSON line (rank 0).  `value` = rays (pixels) per second, whole job.
"""
import argparse
import json
import os
import sys
import time



def _self_launch_before_hip():
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: start the N ranks here, as fresh
    child processes of `python -m torch.distributed.run`, BEFORE this process imports torch or touches HIP
    (a process that has initialised the GPU must never be replaced or forked into ranks), pass rank 0's one
    JSON line through, and exit with the children's status.  Under torchrun (WORLD_SIZE set) this is a no-op."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    n = 1
    for i, arg in enumerate(sys.argv):
        if arg == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif arg.startswith("--gpus="):
            n = int(arg.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool's hosts
    proc = subprocess.run(cmd, env=env)
    sys.exit(proc.returncode)


_self_launch_before_hip()
import numpy as np  # noqa: E402


def _hw_queues_before_hip():
    """A rank's share of an N-rank frame (1/N of the tiles) no longer fills its GPU, so frames
    overlap on separate streams (`frames_in_flight`); HIP maps streams onto GPU_MAX_HW_QUEUES
    hardware queues (4 by default) and launches that share a queue run one after the other.
    Measured on one MI355X with rank 0's share of 8 (--emulate-world 8 --pipeline 8): 4 queues
    0.37 ms per frame, 8: 0.33, 16: 0.25 (an eighth of the whole-frame kernel is 0.24).  The
    runtime reads the variable when it initialises, i.e. it has to be set before the first HIP call."""
    want = 0
    for i, arg in enumerate(sys.argv):
        if arg == "--hw-queues" and i + 1 < len(sys.argv):
            want = int(sys.argv[i + 1])
        elif arg.startswith("--hw-queues="):
            want = int(arg.split("=", 1)[1])
    if want == 0 and (int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--emulate-world" in sys.argv):
        want = 16
    if want > 0:
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(want, 32))


_hw_queues_before_hip()
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sdf_amd as S  # noqa: E402
from sdf_amd import partition as P  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402

VALU_PEAK_UNFUSED_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, one op per lane-issue
VALU_PEAK_FMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (vector)" (counts FMA as 2)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed frames (200 x ~1.7 ms: a third of a second timed)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, help="BASELINE config (2, 3 or 4 are frames)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--bounds", default="auto", choices=["auto", "aabb", "obb", "none", "bvh"])
    ap.add_argument("--skip-ext", type=float, default=0.0, help="SDFGPU_OPT_SKIP_EXTENSION (<0: half the mean primitive size)")
    ap.add_argument("--no-tile-order", action="store_true", help="SDFGPU_OPT_TILE_ORDER=0: tiles in raster order")
    ap.add_argument("--compaction", action="store_true", help="multi-pass march with ray compaction (SDFGPU_OPT_COMPACTION=1)")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="frames in flight (1 = strictly one after the other; 0 = default: 2 on one GPU, "
                         "min(8, N + 2) on N GPUs, where a rank's tile share no longer fills its GPU)")
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="GPU_MAX_HW_QUEUES for this process (0 = leave the environment's value on one GPU, 16 on N > 1 "
                         "GPUs where several frames' launches have to overlap; at most 32)")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="diagnostic (1 GPU): render only rank 0's tile share of an N-rank frame, no gather")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: rehearsal of the N-rank path)")
    ap.add_argument("--device", type=int, default=-1, help="GPU index for every rank (rehearsal: N ranks on one GPU); default LOCAL_RANK")
    ap.add_argument("--lib-comm", action="store_true",
                    help="N > 1: the timed frames go through the library's own communicator (sdfgpu_render_frame_comm: grouped "
                         "ncclSend / ncclRecv to rank 0) instead of torch.distributed's RCCL all-gather.  Until a multi-GPU record of "
                         "that path exists it is measured as a separate, watchdog-guarded leg after the headline (line['lib_comm'])")
    ap.add_argument("--no-lib-comm-leg", action="store_true", help="N > 1: skip the library-communicator leg")
    ap.add_argument("--no-variants", action="store_true", help="skip the variants leg (profiling: kernel averages then cover the timed frames only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-knn", action="store_true", help="skip the kNN (config 5) leg")
    ap.add_argument("--knn-points", type=int, default=10_000_000)
    ap.add_argument("--knn-steps", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU time budget of the baseline leg")
    return ap.parse_args()


def cpu_baseline(buf, cam, mp, w, h, budget_s):
    """Oracle timed on this host's cores on a strided sample of the same frame,
    sized by a short pilot so that the leg costs about `budget_s` seconds."""
    from oracle import oracle as O

    threads = max(1, min(O.hardware_threads(), len(os.sched_getaffinity(0))))
    npix = w * h

    def run(mode, cnt):
        stride = max(1, npix // cnt)
        cnt = min(cnt, (npix + stride - 1) // stride)
        t0 = time.perf_counter()
        O.render(buf, cam, w, h, mp, mode=mode, threads=threads, pix_first=0, pix_count=cnt, pix_stride=stride)   # `threads` as bound at call time
        return cnt, stride, time.perf_counter() - t0

    def sized(mode, pilot, share):
        cnt, _, t = run(mode, pilot)
        want = int(min(npix, max(pilot, cnt / max(t, 1e-6) * budget_s * share)))
        return run(mode, want)

    cnt, stride, t = sized(O.MODE_TUNED, 4096, 0.5)
    out = {
        "value": cnt / t / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{cnt} rays (every {stride}th pixel of the same frame), oracle_render MODE_TUNED "
                  f"(C++ restatement, best-so-far pruning), {threads} threads, {t:.2f} s",
    }
    # the same tuned port on ONE host thread (1 thread and all cores)
    threads_all, threads = threads, 1
    cnt1, stride1, t1 = sized(O.MODE_TUNED, 1024, 0.15)
    threads = threads_all
    out["one_thread_value"] = cnt1 / t1 / 1e6
    out["one_thread_sample"] = f"{cnt1} rays (every {stride1}th pixel), oracle_render MODE_TUNED, 1 thread, {t1:.2f} s"
    cnt_f, stride_f, t_f = sized(O.MODE_FAUX_FAITHFUL, 128, 0.25)
    out["faithful_value"] = cnt_f / t_f / 1e6
    out["faithful_sample"] = (f"{cnt_f} rays (every {stride_f}th pixel), faux_shader::nearest_neighbor as written "
                              f"(O(N) walk, 256-frame allocation per query), {threads} threads, {t_f:.2f} s")
    return out


def knn_leg(a, rank, world, local_rank, dev, dist):
    """BASELINE config 5: k = 8 nearest primitives for 10 M points against the
    100k-primitive tree; the point array is cut into contiguous slices, one per
    rank, results stay sharded (no collective on the data path)."""
    wl = W.CONFIGS[5]
    buf = W.scene(wl)
    gpu = S.SdfGpu(local_rank)
    info = gpu.upload_tree(buf)
    n_total = a.knn_points
    lo, hi = P.query_slice(rank, world, n_total)
    pts_all = W.query_points(wl, n_total)
    pts = torch.from_numpy(pts_all[lo:hi]).to(dev)
    n = hi - lo
    k = wl.k
    d_out = torch.empty(n * k, dtype=torch.float32, device=dev)
    i_out = torch.empty(n * k, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        gpu.knn_dev(pts.data_ptr(), n, k, d_out.data_ptr(), i_out.data_ptr(), stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.knn_steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    # the k = 1 query (faux_shader::nearest_neighbor itself) on the same points and tree
    step1 = lambda: gpu.nearest_dev(pts.data_ptr(), n, d_out.data_ptr(), i_out.data_ptr(), stream)  # noqa: E731
    step1()
    barrier()
    t1 = time.perf_counter()
    for _ in range(a.knn_steps):
        step1()
    barrier()
    el1 = time.perf_counter() - t1
    if dist is not None:
        t = torch.tensor([el1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el1 = float(t.item())
    def timed(fn):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.knn_steps):
            fn()
        barrier()
        e = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([e], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        return e

    out = {
        "metric": "kNN queries/s (k=8)", "value": n_total * a.knn_steps / el, "unit": "queries/s",
        "nearest_k1": {"value": n_total * a.knn_steps / el1, "unit": "queries/s", "ms_per_step": el1 / a.knn_steps * 1e3},
        "ms_per_step": el / a.knn_steps * 1e3, "steps": a.knn_steps,
        "config": {"workload": wl.name, "n_points": n_total, "k": k, "n_prims": wl.n_prims, "seed": hex(wl.seed),
                   "tree_levels": int(info.max_level), "tree_bytes_hbm": int(info.device_bytes),
                   "points": "uniform in 1.2x the scene cube, unsorted (reordered along the Hilbert curve inside the call)",
                   "partition": "contiguous query slices, results sharded"},
    }
    # the same batch handed over already sorted along a space-filling curve by the caller -- with the library's own
    # reordering still on (it sorts a sorted batch) and with it off (the walk alone: no permutation traffic)
    pts_sorted = torch.from_numpy(W.query_points(wl, n_total, morton=True)[lo:hi]).to(dev)
    step_s = lambda: gpu.knn_dev(pts_sorted.data_ptr(), n, k, d_out.data_ptr(), i_out.data_ptr(), stream)  # noqa: E731
    step1_s = lambda: gpu.nearest_dev(pts_sorted.data_ptr(), n, d_out.data_ptr(), i_out.data_ptr(), stream)  # noqa: E731
    e_on, e1_on = timed(step_s), timed(step1_s)
    gpu.set_reorder(False)
    e_off, e1_off = timed(step_s), timed(step1_s)
    gpu.set_reorder(True)
    # two batches in flight: consecutive batches alternate between two streams of the same context (each
    # stream has its own sort buffers and its own outputs here), so the bandwidth-bound sort of one batch
    # runs beside the VALU-bound walk of the other -- the query-side analogue of the frame leg's frames in flight
    s2 = [torch.cuda.Stream(device=dev) for _ in range(2)]
    o2 = [(torch.empty(n * k, dtype=torch.float32, device=dev), torch.empty(n * k, dtype=torch.int32, device=dev)) for _ in range(2)]
    flip = [0]

    def step_two():
        j = flip[0] & 1
        flip[0] += 1
        gpu.knn_dev(pts.data_ptr(), n, k, o2[j][0].data_ptr(), o2[j][1].data_ptr(), s2[j].cuda_stream)

    e_two = timed(step_two)
    out["two_batches_in_flight"] = {"value": n_total * a.knn_steps / e_two, "ms_per_step": e_two / a.knn_steps * 1e3, "unit": "queries/s",
                                    "note": "the same unsorted batch, consecutive calls alternating between two streams of one context"}
    out["presorted"] = {
        "points": "the same points, sorted along the Hilbert curve by the caller before the call (workloads.hilbert_keys)",
        "k8_reorder_on": {"value": n_total * a.knn_steps / e_on, "ms_per_step": e_on / a.knn_steps * 1e3},
        "k8_reorder_off": {"value": n_total * a.knn_steps / e_off, "ms_per_step": e_off / a.knn_steps * 1e3},
        "k1_reorder_on": {"value": n_total * a.knn_steps / e1_on, "ms_per_step": e1_on / a.knn_steps * 1e3},
        "k1_reorder_off": {"value": n_total * a.knn_steps / e1_off, "ms_per_step": e1_off / a.knn_steps * 1e3},
        "unit": "queries/s",
    }
    if rank == 0 and world == 1:
        # roofline of the kNN kernel: units from the counting instantiation, duration from HIP events
        # around isolated launches with the permutation already made (the kernel alone)
        st = gpu.knn_stats_dev(pts.data_ptr(), n, k)
        kms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            torch.cuda.synchronize()
            kms.append(e0.elapsed_time(e1))
        call_ms = float(np.mean(kms))
        tr, trn = profiled_traffic("knn_kernel<1, 8, false")
        whole_call = None
        try:   # FETCH_SIZE + WRITE_SIZE of every kernel of one unsorted k = 8 call (tools/query_traffic.sh, committed under profiles/)
            import glob
            qt = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "query_traffic.json")))
            if qt:
                with open(qt[-1]) as f:
                    row = json.load(f)["unsorted_k8"]
                whole_call = {"fetched_MB": row["fetched_MB"], "written_MB": row["written_MB"], "algorithmic_MB": row["algorithmic_MB"],
                              "ratio": row["ratio"], "source": os.path.relpath(qt[-1], ROOT),
                              "note": "all kernels of one call (keys, sort, walk); the excess over 1 is the random permutation of 10 M small "
                                      "items at the memory system's granularity (64-byte fetches for 16-byte points, 32-byte sectors for "
                                      "8-byte sort pairs), not re-reads: the caller-sorted call moves 1.1x"}
        except (OSError, ValueError, KeyError):
            whole_call = None
        out["roofline"] = roofline_object(st, call_ms, "sdfgpu::knn_kernel<1, 8, false, false>", n * 16 + n * k * 8 + int(info.device_bytes),
                                          note="kernel_ms_avg = the whole sdfgpu_knn_dev call (curve keys + radix sort + walk) under HIP "
                                               "events, so the kernel's own rate is higher (profiles/ has its rocprofv3 duration); fp32 "
                                               "VALU-bound walk, K-slot insertion counted as 2 flops per offered primitive",
                                          traffic=tr, traffic_note=trn)
        out["roofline"]["whole_call_traffic"] = whole_call
        out["roofline"]["valu_issue"] = issue_utilisation("knn_kernel<1, 8, false", call_ms)   # of the whole call (sort included)
        out["work"] = {"inner_visits_per_query": st["inner_visits"] / max(1, st["queries"]),
                       "inner_needed_per_query": st["inner_needed"] / max(1, st["queries"]),
                       "leaf_evals_per_query": (st["sphere_evals"] + st["box_evals"]) / max(1, st["queries"]),
                       "leaf_needed_per_query": (st["sphere_needed"] + st["box_needed"]) / max(1, st["queries"]),
                       "wave_inner_visits": st["wave_inner"], "wave_leaf_visits": st["wave_leaf"]}
    if rank == 0 and world == 1 and not a.no_cpu:
        from oracle import oracle as O
        threads = max(1, min(O.hardware_threads(), len(os.sched_getaffinity(0))))

        def cpu_knn(nthreads, budget_s):
            m = 500
            t0 = time.perf_counter()
            O.knn(buf, pts_all[:m], k, threads=nthreads, tuned=True)
            tp = time.perf_counter() - t0
            m = int(min(n_total, max(m, m / max(tp, 1e-6) * budget_s)))
            t0 = time.perf_counter()
            O.knn(buf, pts_all[:m], k, threads=nthreads, tuned=True)
            tc = time.perf_counter() - t0
            return m, tc

        m1, t1c = cpu_knn(1, 3.0)
        mN, tNc = cpu_knn(threads, 4.0)
        out["cpu_baseline"] = {"value": mN / tNc, "unit": "queries/s", "cores": threads, "kind": "port",
                               "sample": f"k=8, {mN} of the same points, oracle_knn_tuned (C++ restatement, pruned against the k-th best), "
                                         f"{threads} threads, {tNc:.2f} s",
                               "one_thread_value": m1 / t1c,
                               "one_thread_sample": f"k=8, {m1} of the same points, 1 thread, {t1c:.2f} s"}
    gpu.close()
    return out


def profiled_traffic(kernel_substr="render_pass_kernel<1, false, true, false"):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE are collected in separate runs by tools/profile_bench.sh;
    bench.py itself cannot read counters).  FETCH_SIZE/WRITE_SIZE are in KiB; the reads of
    this kernel are 64-byte scalar-cache lines, not the wide streaming reads whose FETCH_SIZE
    gfx950 halves, so no doubling is applied (MI355X_MICROARCH.md, HBM section)."""
    import glob
    import json as _json
    best = None
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "pmc_by_kernel.json"))):
        try:
            with open(path) as f:
                d = _json.load(f)
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            # exactly this instantiation (profiles of different rounds name it with or without the trailing CHAIN = false)
            if (k.endswith(kernel_substr + ">") or k.endswith(kernel_substr + ", false>")) and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                best = (path, (v["FETCH_SIZE"]["mean_per_launch"] + v["WRITE_SIZE"]["mean_per_launch"]) * 1024.0)
    if best is None:
        return None, "no PMC profile committed"
    return best[1], "bytes per launch, FETCH_SIZE + WRITE_SIZE from " + os.path.relpath(best[0], os.path.dirname(os.path.abspath(__file__)))


def profiled_counter(kernel_substr, counter):
    """Mean per launch of one committed PMC counter of exactly this instantiation (latest profile), or None."""
    import glob
    import json as _json
    val = None
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "pmc_by_kernel.json"))):
        try:
            with open(path) as f:
                d = _json.load(f)
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            if (k.endswith(kernel_substr + ">") or k.endswith(kernel_substr + ", false>")) and counter in v:
                val = (v[counter]["mean_per_launch"], os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))))
    return val


VALU_NS_PER_INST_PER_SIMD = 1.09   # tools/ubench/issue_rates.hip: a SIMD with 8 resident wavefronts retires one plain fp32 VALU instruction per 1.09 ns
N_SIMDS = 1024


def issue_utilisation(kernel_substr, kernel_ms):
    """How busy the kernel keeps the vector pipes: SQ_INSTS_VALU (committed PMC pass) x the measured issue time of one
    instruction / SIMDs / kernel time.  This -- not flops -- is the wall these walks run into."""
    c = profiled_counter(kernel_substr, "SQ_INSTS_VALU")
    if c is None or kernel_ms <= 0:
        return None
    busy_ms = c[0] * VALU_NS_PER_INST_PER_SIMD * 1e-6 / N_SIMDS
    return {"valu_instructions_per_launch": c[0], "ns_per_instruction_per_simd": VALU_NS_PER_INST_PER_SIMD, "simds": N_SIMDS,
            "issue_bound_ms": busy_ms, "frac_of_kernel_time": busy_ms / kernel_ms, "source": c[1],
            "note": "wave-level vector instructions x measured issue time (tools/ubench/issue_rates.hip) / 1024 SIMDs: the time the "
                    "kernel would take if every SIMD issued back to back"}


def roofline_object(st, kernel_ms, kernel_name, bytes_alg, note, traffic=None, traffic_note=None):
    """The `roofline` object of the JSON line for a VALU-bound walk kernel.  `achieved` counts the units
    a lane itself needed (its own bound asked for the node / primitive); `achieved_issued` also counts the
    lanes that only rode along on a node their wavefront visited for somebody else."""
    f_need, f_issued = W.algorithmic_flops_needed(st), W.algorithmic_flops(st)
    t = kernel_ms * 1e-3
    ach = f_need / t / 1e12
    return {
        "bound": "valu",
        "achieved": ach,
        "peak": VALU_PEAK_FMA_TFLOPS,
        "unit": "TFLOP/s",
        "frac": ach / VALU_PEAK_FMA_TFLOPS,
        "traffic": traffic,
        "traffic_note": traffic_note,
        "kernel": kernel_name,
        "kernel_ms_avg": kernel_ms,
        "algorithmic_flops_per_launch": f_need,
        "peak_note": "157.3 = MI355X_MICROARCH.md peak fp32 vector (FMA counted as 2); leaf arithmetic must stay unfused "
                     "for bit parity, so one rounded op per lane-issue = 78.6 TFLOP/s is the ceiling this code can approach",
        "frac_of_unfused_peak": ach / VALU_PEAK_UNFUSED_TFLOPS,
        "achieved_issued": f_issued / t / 1e12,
        "frac_issued": f_issued / t / 1e12 / VALU_PEAK_FMA_TFLOPS,
        "lane_utility": f_need / max(1, f_issued),
        "hbm": {"algorithmic_bytes_per_launch": bytes_alg, "achieved_GBs": bytes_alg / t / 1e9, "peak_GBs": HBM_PEAK_GBS,
                "frac": bytes_alg / t / 1e9 / HBM_PEAK_GBS},
        "note": note,
    }


def variants(a, gpu, cam, mp, w, h, slots, depth, wl, dev):
    """The same frame under the conditions the headline does not cover (N = 1, outside the timed region):
    strictly one frame after the other, a moving camera (the learned tile order is always stale), tile
    ordering off, the very first frame of a fresh context, and tree upload times."""
    import math
    npix = w * h

    def run(step, frames):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for kk in range(frames):
            step(kk)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3

    def frame(g, c, sl):
        g.render_dev(c, w, h, mp, sl.o_hit.data_ptr(), sl.o_depth.data_ptr(), sl.o_normal.data_ptr(), sl.o_prim.data_ptr(),
                     sl.stream.cuda_stream)

    out = {}
    frames = max(4, a.steps)
    ms1 = run(lambda kk: frame(gpu, cam, slots[0]), frames)
    out["pipeline1"] = {"value": npix / ms1 / 1e3, "ms_per_step": ms1, "note": "one frame in flight, one stream"}
    L = S.default_extent(wl.n_prims)
    cams = [S.camera_look_at((2.5 * L * math.sin(0.02 * i), 0.0, -2.5 * L * math.cos(0.02 * i)), (0, 0, 0), (0, 1, 0),
                             math.pi / 4, w, h) for i in range(frames)]
    msm = run(lambda kk: frame(gpu, cams[kk], slots[kk % depth]), frames)
    out["moving_camera"] = {"value": npix / msm / 1e3, "ms_per_step": msm,
                            "note": f"orbit, 0.02 rad per frame, {depth} frames in flight; tile order learned from a different view"}
    gpu.set_tile_order(False)
    mso = run(lambda kk: frame(gpu, cam, slots[kk % depth]), frames)
    gpu.set_tile_order(not a.no_tile_order)
    out["tile_order_off"] = {"value": npix / mso / 1e3, "ms_per_step": mso, "note": f"tiles in raster order, {depth} frames in flight"}
    # the same scene from 60 scene extents away (field of view narrowed to keep it in frame): beyond the range of
    # the scaled box test, served by the far version of the march kernel ("Far walk")
    far_cam = S.camera_look_at((0.0, 0.0, -60.0 * L), (0, 0, 0), (0, 1, 0), math.pi / 4 * 2.5 / 60.0, w, h)
    far_mp = S.march_params(max_steps=int(mp.max_steps), eps_hit=float(mp.eps_hit), t_min=0.0, t_max=64.0 * L, normal_h=float(mp.normal_h))

    def far_frame(kk):
        sl = slots[kk % depth]
        gpu.render_dev(far_cam, w, h, far_mp, sl.o_hit.data_ptr(), sl.o_depth.data_ptr(), sl.o_normal.data_ptr(), sl.o_prim.data_ptr(),
                       sl.stream.cuda_stream)

    for kk in range(2 * depth):
        far_frame(kk)
    msf = run(far_frame, min(frames, 50))
    out["far_camera"] = {"value": npix / msf / 1e3, "ms_per_step": msf,
                         "note": f"eye 60 scene extents away, {depth} frames in flight (1.7 ms with the far walk; 10 ms before it existed)"}
    for kk in range(2 * depth):   # leave every stream's tile schedule as the headline camera's
        frame(gpu, cam, slots[kk % depth])
    torch.cuda.synchronize()
    # cold: fresh context, tree just uploaded, first frame (no schedule, buffers of the context untouched)
    g2 = S.SdfGpu(dev.index)
    buf = W.scene(wl)
    t0 = time.perf_counter()
    g2.upload_tree(buf)
    up = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frame(g2, cam, slots[0])
    torch.cuda.synchronize()
    out["cold_frame_ms"] = (time.perf_counter() - t0) * 1e3
    g2.close()
    # the other single-GPU frames of BASELINE.json (configs 2 and 4), same code, a few frames each, two in flight
    for cfg in (2, 4):
        wlx = W.CONFIGS[cfg]
        wx, hx = wlx.width, wlx.height
        gx = S.SdfGpu(dev.index)
        infx = gx.upload_tree(W.scene(wlx))
        camx, mpx = W.camera(wlx, wx, hx), W.march(wlx)
        outs = [(torch.empty(wx * hx, dtype=torch.uint8, device=dev), torch.empty(wx * hx, dtype=torch.float32, device=dev),
                 torch.empty(wx * hx * 3, dtype=torch.float32, device=dev), torch.empty(wx * hx, dtype=torch.int32, device=dev))
                for _ in range(depth)]

        def framex(kk):
            o = outs[kk % depth]
            gx.render_dev(camx, wx, hx, mpx, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), slots[kk % depth].stream.cuda_stream)

        for kk in range(2 * depth):
            framex(kk)
        nfr = 40 if cfg == 2 else 12
        msx = run(framex, nfr)
        out[f"config{cfg}"] = {"value": wx * hx / msx / 1e3, "ms_per_step": msx, "unit": "Mrays/s", "frames": nfr,
                               "note": f"{wlx.name}: {wx}x{hx}, {wlx.n_prims} primitives, one MI355X, {depth} frames in flight; "
                                       f"resident tree + grid {infx.device_bytes / 1e6:.1f} MB ({int(infx.grid_cells)} grid cells)"}
        del outs
        gx.close()
    # The same scene as the crate's own builder emits it: every merged box carries its PCA trans_inverse (obb.rs:183-185),
    # so every level re-bases the point (faux_shader.rs:165,175).  Pruned since round 3 ("chain path": the library's
    # hierarchy over the composed maps' world boxes, leaf values through the reference's one rounded mat-vec per level).
    crate = S.random_scene(wl.n_prims, wl.seed, opts={"pca_trans_inverse": True}).expanded().make_buffer()
    gc = S.SdfGpu(dev.index)
    infc = gc.upload_tree(crate)

    def crate_frame(kk):
        sl = slots[kk % depth]
        gc.render_dev(cam, w, h, mp, sl.o_hit.data_ptr(), sl.o_depth.data_ptr(), sl.o_normal.data_ptr(), sl.o_prim.data_ptr(), sl.stream.cuda_stream)

    for kk in range(2 * depth):
        crate_frame(kk)
    msc = run(crate_frame, 20)
    out["crate_buffer"] = {"value": npix / msc / 1e3, "ms_per_step": msc, "unit": "Mrays/s", "frames": 20,
                           "ratio_to_identity_buffer_frame": msc / ms1,
                           "bounds": {1: "aabb", 2: "obb", 3: "none", 4: "bvh"}[int(infc.bounds)], "identity_inner": int(infc.identity_inner),
                           "note": f"{wl.name} built with pca_trans_inverse (the crate's literal merged boxes): chain path, {depth} frames in "
                                   "flight; before round 3 such a buffer was walked exhaustively, O(nodes) per query"}
    gc.close()
    # The headline scene with every union's smooth radius set (elements.rs:207-211) and the SMOOTH_UNION extension on:
    # the general path (post-order combine over the buffer's own topology), whose walks prune since round 3
    # (walk_eval_pruned: cuts).  Its march has none of the pruned path's machinery, so the frame is half the
    # headline's size each way; the exhaustive walk of the same tree (SDFGPU_GEN_PRUNE=0) is timed beside it on a small frame.
    buf_h = W.scene(wl)
    sm = S.SdfTreeBuffer(buf_h.downtree_buffer, buf_h.uptree_buffer.copy())
    dtb = sm.downtree_buffer
    idx = np.arange(len(dtb), dtype=np.int64)
    post = idx + dtb["len"].astype(np.int64) - (dtb["level"].astype(np.int64) - 1)   # post-order position of pre-order node i
    unions = (dtb["is_primitive"] == 0) & (dtb["op_code"] == 0)
    k_smooth = 1.0
    fl = sm.uptree_buffer["op_specific"]["floats"]
    fl[post[unions], 0] = k_smooth
    res = {}
    for tag, env, div in (("pruned_small", None, 8), ("exhaustive_small", "0", 8), ("pruned", None, 2)):
        ws, hs = max(8, w // div), max(8, h // div)
        cams = W.camera(wl, ws, hs)
        os_ = (torch.empty(ws * hs, dtype=torch.uint8, device=dev), torch.empty(ws * hs, dtype=torch.float32, device=dev),
               torch.empty(ws * hs * 3, dtype=torch.float32, device=dev), torch.empty(ws * hs, dtype=torch.int32, device=dev))
        if env is None:
            os.environ.pop("SDFGPU_GEN_PRUNE", None)
        else:
            os.environ["SDFGPU_GEN_PRUNE"] = env
        gs = S.SdfGpu(dev.index)
        infs = gs.upload_tree(sm, extensions=S.EXT_SMOOTH_UNION)
        st_ = torch.cuda.current_stream().cuda_stream

        def sframe(kk):
            gs.render_dev(cams, ws, hs, mp, os_[0].data_ptr(), os_[1].data_ptr(), os_[2].data_ptr(), os_[3].data_ptr(), st_)

        sframe(0)
        mss = run(sframe, 1 if env is not None else 2)
        res[tag] = (mss, int(infs.walk), int(os_[0].sum().item()), ws, hs)
        gs.close()
        del os_
    os.environ.pop("SDFGPU_GEN_PRUNE", None)
    pr, ps, ex = res["pruned"], res["pruned_small"], res["exhaustive_small"]
    out["smooth_union"] = {"value": pr[3] * pr[4] / pr[0] / 1e3, "ms_per_step": pr[0], "unit": "Mrays/s",
                           "width": pr[3], "height": pr[4], "smooth_radius": k_smooth, "walk": pr[1],
                           "small_frame": {"width": ps[3], "height": ps[4], "pruned_ms": ps[0], "exhaustive_ms": ex[0],
                                           "speedup_over_exhaustive": ex[0] / ps[0], "same_hit_count": ps[2] == ex[2]},
                           "note": f"{wl.name} with smooth radius {k_smooth} on every union, SDFGPU_EXT_SMOOTH_UNION: general path, pruned by cuts "
                                   "(walk 2 = SDFGPU_WALK_GENERAL_PRUNED), one frame in flight, half the headline's size each way; the "
                                   "exhaustive walk of the same tree is timed on a frame an eighth of the size each way (506 wavefronts: "
                                   "neither fills the GPU there)"}
    for kk in range(2 * depth):   # slots back to the headline scene's frame
        frame(gpu, cam, slots[kk % depth])
    torch.cuda.synchronize()
    # tree flatten (host builder's make_buffer) and upload (validate + SAH build + copy), per scene size
    out["upload_ms"] = {}
    for cfg in (2, 3, 4):
        wlx = W.CONFIGS[cfg]
        node = S.random_scene(wlx.n_prims, wlx.seed)
        fl, upx = [], []
        for _ in range(2):   # the better of two: the first upload of a process also loads the grid build's code objects
            t0 = time.perf_counter()
            b = node.expanded().make_buffer()
            fl.append((time.perf_counter() - t0) * 1e3)
            g3 = S.SdfGpu(dev.index)
            t0 = time.perf_counter()
            g3.upload_tree(b)
            torch.cuda.synchronize()
            upx.append((time.perf_counter() - t0) * 1e3)
            g3.close()
        out["upload_ms"][f"{wlx.n_prims}_prims"] = {"flatten_expanded_make_buffer_ms": min(fl), "sdfgpu_upload_tree_ms": min(upx),
                                                    "note": "better of two runs; upload = validate + SAH build + candidate grid + copies"}
    return out


def make_lib_comm(gpu, rank, world, dist, dev, cam, w, h, mp, slots):
    """The library's communicator over the ranks of this job, or None (on every rank alike) when it cannot be
    made or its trial frame reports an error."""
    comm = None
    try:
        box = [S.SdfComm.make_id(gpu) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = S.SdfComm(gpu, box[0], rank, world)
    except Exception as e:   # noqa: BLE001
        print(f"[bench rank {rank}] library communicator unavailable ({e})", file=sys.stderr)
        comm = None
    ok = torch.tensor([1 if comm is not None else 0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if comm is not None:
            comm.close()
        return None
    good = 1
    try:   # one trial frame: an error it reports (as opposed to a hang) sends every rank back to the torch path
        sl = slots[0]
        comm.render_frame(cam, w, h, mp, 0, *(sl.out_ptrs if rank == 0 else (0, 0, 0, 0)), sl.stream.cuda_stream)
        torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001
        print(f"[bench rank {rank}] library communicator failed on its trial frame ({e})", file=sys.stderr)
        good = 0
    ok = torch.tensor([good], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        comm.close()
        return None
    return comm


def lib_comm_leg(a, gpu, rank, world, dist, dev, cam, w, h, mp, slots, depth, line, knn):
    """N > 1, last thing in the run: the same frames through sdfgpu_render_frame_comm.  The path is unexercised on
    real multi-GPU hardware, and a collective that goes wrong hangs a stream rather than raising, so a watchdog
    thread prints the line WITHOUT this leg's numbers and ends the process if the leg does not finish in time."""
    import threading
    done = threading.Event()
    limit = float(os.environ.get("SDF_BENCH_LIBCOMM_TIMEOUT", "90"))

    def watchdog():
        if done.wait(limit):
            return
        if rank == 0:
            line["lib_comm"] = {"status": f"did not finish within {limit:.0f} s (hang in the exchange?); headline unaffected"}
            if knn is not None:
                line["knn"] = knn
            print(json.dumps(line), flush=True)
        os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    out = {"status": "communicator could not be made or failed its trial frame"}
    comm = make_lib_comm(gpu, rank, world, dist, dev, cam, w, h, mp, slots)
    if comm is not None:
        npix = w * h
        ref = None
        if rank == 0:   # the frame the torch path assembled (slot 0 holds it), to compare with
            sl = slots[0]
            ref = [t.clone() for t in (sl.o_hit, sl.o_depth, sl.o_normal, sl.o_prim)]

        def step(k):
            sl = slots[k % depth]
            comm.render_frame(cam, w, h, mp, 0, *(sl.out_ptrs if rank == 0 else (0, 0, 0, 0)), sl.stream.cuda_stream)

        for k in range(2 * depth):
            step(k)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step(k)
        dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        out = {"status": "ok", "value": npix * a.steps / el / 1e6, "unit": "Mrays/s", "ms_per_step": el / a.steps * 1e3,
               "gather": "RCCL grouped ncclSend / ncclRecv to rank 0 + untile (sdfgpu_render_frame_comm)"}
        if rank == 0:
            sl = slots[0]
            same = (torch.equal(ref[0], sl.o_hit) and torch.equal(ref[3], sl.o_prim)
                    and torch.equal(ref[1].view(torch.int32), sl.o_depth.view(torch.int32))
                    and torch.equal(ref[2].view(torch.int32), sl.o_normal.view(torch.int32)))
            out["assembly_check"] = "bit-identical to the torch all-gather frame" if same else "MISMATCH"
        comm.close()
    done.set()
    return out


def trace(msg):
    """Stage markers for rehearsing the N-rank path (SDF_BENCH_TRACE=1)."""
    if os.environ.get("SDF_BENCH_TRACE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.device >= 0:
        local_rank = a.device
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:   # (--gpus N without a launcher never gets here: _self_launch_before_hip started the ranks)
        a.gpus = world
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = W.CONFIGS[a.config]
    w, h = a.width or wl.width, a.height or wl.height
    buf = W.scene(wl)
    cam, mp = W.camera(wl, w, h), W.march(wl)
    bounds = {"auto": S.BOUNDS_AUTO, "aabb": S.BOUNDS_AABB, "obb": S.BOUNDS_OBB, "none": S.BOUNDS_NONE, "bvh": S.BOUNDS_BVH}[a.bounds]
    gpu = S.SdfGpu(local_rank)
    info = gpu.upload_tree(buf, bounds=bounds)
    gpu.set_skip_extension(a.skip_ext)
    gpu.set_compaction(a.compaction)
    gpu.set_tile_order(not a.no_tile_order)

    npix = w * h
    tiles = P.n_tiles(w, h)
    # Frame pipelining: consecutive frames alternate between `depth` slots, each with its own
    # stream and output buffers, so the drain of one frame (a few long grazing rays on a mostly
    # idle GPU) overlaps the start of the next.  Frames are independent; each is complete
    # (march + normals [+ gather + assembly]) before its slot is reused.
    depth = a.pipeline if a.pipeline > 0 else (2 if world == 1 else min(8, world + 2))
    main_stream = torch.cuda.current_stream()
    streams = [main_stream] + [torch.cuda.Stream(device=dev) for _ in range(depth - 1)]

    class Slot:
        pass

    slots = []
    for sidx in range(depth):
        sl = Slot()
        sl.stream = streams[sidx]
        sl.o_hit = torch.empty(npix, dtype=torch.uint8, device=dev)
        sl.o_depth = torch.empty(npix, dtype=torch.float32, device=dev)
        sl.o_normal = torch.empty(npix * 3, dtype=torch.float32, device=dev)
        sl.o_prim = torch.empty(npix, dtype=torch.int32, device=dev)
        if world > 1:
            nbytes = P.packed_bytes(world, tiles)
            # one packed byte buffer per rank: [depth f32 | normal 3xf32 | prim u32 | hit u8]
            sl.mine = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            sl.gathered = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
        slots.append(sl)
    o_hit = slots[0].o_hit
    torch.cuda.synchronize()

    if world == 1 and a.emulate_world > 1:
        # diagnostic only: rank 0's share of an N-rank frame (no gather), to size the pipeline depth
        first, stride, cnt = P.rank_tiles(0, a.emulate_world, tiles)

        def step(k=0):
            sl = slots[k % depth]
            gpu.render_tiles_dev(cam, w, h, mp, first, stride, cnt, sl.o_hit.data_ptr(), sl.o_depth.data_ptr(),
                                 sl.o_normal.data_ptr(), sl.o_prim.data_ptr(), sl.stream.cuda_stream)
    elif world == 1:
        def step(k=0):
            sl = slots[k % depth]
            gpu.render_dev(cam, w, h, mp, sl.o_hit.data_ptr(), sl.o_depth.data_ptr(), sl.o_normal.data_ptr(),
                           sl.o_prim.data_ptr(), sl.stream.cuda_stream)
    else:
        first, stride, cnt = P.rank_tiles(rank, world, tiles)
        nbytes = P.packed_bytes(world, tiles)

        # device addresses of every plane, once (the buffers live as long as the slots)
        for sl in slots:
            sl.mine_ptrs = tuple(t.data_ptr() for t in P.planes(sl.mine, world, tiles))
            sl.out_ptrs = (sl.o_hit.data_ptr(), sl.o_depth.data_ptr(), sl.o_normal.data_ptr(), sl.o_prim.data_ptr())

        # Assembly.  Default: torch.distributed's RCCL all-gather of the packed shares + one untile launch on rank 0.
        # --lib-comm: the library's own communicator (sdfgpu_render_frame_comm: tile share -> grouped ncclSend /
        # ncclRecv to rank 0 -> untile), the path a host without torch uses.  That path has never run on more than
        # one GPU, so by default it is a separate leg after the headline, guarded by a watchdog (lib_comm_leg).
        comm = make_lib_comm(gpu, rank, world, dist, dev, cam, w, h, mp, slots) if (a.lib_comm and a.backend == "nccl") else None
        gather_kind = "RCCL grouped send/recv to rank 0 (sdfgpu_render_frame_comm)" if comm is not None else (f"{a.backend} all-gather (torch.distributed)")

        def step(k=0):
            sl = slots[k % depth]
            if comm is not None:
                ptrs = sl.out_ptrs if rank == 0 else (0, 0, 0, 0)
                comm.render_frame(cam, w, h, mp, 0, *ptrs, sl.stream.cuda_stream)
                return
            with torch.cuda.stream(sl.stream):
                gpu.render_tiles_dev(cam, w, h, mp, first, stride, cnt, *sl.mine_ptrs, sl.stream.cuda_stream)
                if a.backend == "nccl":
                    dist.all_gather_into_tensor(sl.gathered, sl.mine)
                else:   # rehearsal on a CPU backend: stage through host memory (synchronous)
                    sl.stream.synchronize()
                    host = torch.empty(world * nbytes, dtype=torch.uint8)
                    dist.all_gather_into_tensor(host, sl.mine.cpu())
                    sl.gathered.copy_(host)
                if rank == 0:   # every rank's share in one launch (rank 0's own chunk is the first of the gathered ones)
                    gpu.untile_gathered_dev(w, h, world, 0, sl.gathered.data_ptr(), sl.gathered.data_ptr(), *sl.out_ptrs,
                                            sl.stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # set-up: every slot renders twice, so that each stream has its buffers touched and its tile
    # schedule (recorded by the first frame on that stream) before the W warm-up steps
    for k in range(2 * depth):
        step(k)
    barrier()
    trace("setup done")
    for k in range(a.warmup):
        step(k)
        trace(f"warmup step {k} enqueued")
    barrier()
    trace("warmup done")
    # Duration of the dominant kernel when it has the GPU to itself (roofline): HIP events
    # around single launches on the launch stream, outside the timed region.
    kms = []
    if world == 1:
        for _ in range(min(10, max(3, a.steps))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
            step(0)
            e1.record(main_stream)
            torch.cuda.synchronize()
            kms.append(e0.elapsed_time(e1))
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0
    trace("timed loop done")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    assembly_check = None
    if world > 1 and rank == 0:
        # the frame assembled from every rank's tiles must equal rank 0's own whole-frame render
        chk = [torch.empty_like(t) for t in (slots[0].o_hit, slots[0].o_depth, slots[0].o_normal, slots[0].o_prim)]
        gpu.render_dev(cam, w, h, mp, chk[0].data_ptr(), chk[1].data_ptr(), chk[2].data_ptr(), chk[3].data_ptr(),
                       main_stream.cuda_stream)
        torch.cuda.synchronize()
        same = (torch.equal(chk[0], slots[0].o_hit) and torch.equal(chk[3], slots[0].o_prim)
                and torch.equal(chk[1].view(torch.int32), slots[0].o_depth.view(torch.int32))
                and torch.equal(chk[2].view(torch.int32), slots[0].o_normal.view(torch.int32)))
        assembly_check = "bit-identical to a single-GPU render" if same else "MISMATCH"

    line = None
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        value = npix * a.steps / elapsed / 1e6
        hit_frac = float(o_hit.float().mean().item())
        line = {
            "metric": "Mrays/s, sphere trace + normals",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl.name,
                "width": w, "height": h, "n_prims": wl.n_prims, "seed": hex(wl.seed),
                "tree_nodes": int(info.n_nodes), "tree_levels": int(info.max_level),
                "bounds": {1: "aabb", 2: "obb", 3: "none", 4: "bvh"}[int(info.bounds)],
                "tree_bytes_hbm": int(info.device_bytes),
                "max_steps": int(mp.max_steps), "eps_hit": float(mp.eps_hit), "t_max": float(mp.t_max),
                "normal_h": float(mp.normal_h),
                "frames_in_flight": depth,
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
                "tile_order": not a.no_tile_order,
                "camera": "static (every frame the same view; the longest-first tile order is learned from the previous frame)",
                "partition": "whole frame" if world == 1 else f"8x8 tiles round-robin over {world} ranks + {gather_kind}",
                "hit_fraction": hit_frac,
                "assembly_check": assembly_check,
            },
        }
        if world == 1:
            k_avg = float(np.mean(kms))
            st = gpu.render_stats(cam, w, h, mp)
            bytes_alg = npix * 21 + int(info.device_bytes)
            traffic, traffic_note = profiled_traffic()
            line["roofline"] = roofline_object(
                st, k_avg, "sdfgpu::render_pass_kernel<1, false, true, false, false>", bytes_alg,
                note="fp32 VALU-bound tree walk, no dense contraction (no MFMA); kernel_ms_avg = isolated launches under "
                     "HIP events on the launch stream, outside the timed region",
                traffic=traffic, traffic_note=traffic_note)
            line["roofline"]["valu_issue"] = issue_utilisation("render_pass_kernel<1, false, true, false", k_avg)
            line["roofline"]["frac_at_throughput"] = W.algorithmic_flops_needed(st) / (ms * 1e-3) / 1e12 / VALU_PEAK_FMA_TFLOPS
            if int(info.grid_cells) > 0:
                # The candidate grid (round 3) removed most of the algorithm's own work, so `frac` fell while the frame
                # got faster.  For continuity with rounds 1-2: the same frame's work as the walk-only algorithm does it
                # (counting kernel of a context without the grid) over THIS kernel's time.  Not the kernel's roofline.
                g0 = S.SdfGpu(local_rank)
                g0.set_candidate_grid(False)
                g0.upload_tree(buf, bounds=bounds)
                st0 = g0.render_stats(cam, w, h, mp)
                g0.close()
                f0 = W.algorithmic_flops_needed(st0)
                line["roofline"]["frac_at_round2_work"] = f0 / (k_avg * 1e-3) / 1e12 / VALU_PEAK_FMA_TFLOPS
                line["roofline"]["round2_work_note"] = (f"{f0 / 1e9:.2f} Gflop = needed-lane work of the same frame without the candidate grid "
                                                        "(walks on every uncertified step, as in rounds 1-2) / kernel_ms_avg; continuity only")
            line["work"] = {
                "steps_per_ray": st["steps"] / max(1, st["rays"]),
                "walks_per_ray": st["queries"] / max(1, st["rays"]),
                "wave_steps": st["wave_steps"], "wave_walks": st["wave_walks"],
                "inner_visits_per_walk": st["inner_visits"] / max(1, st["queries"]),
                "inner_needed_per_walk": st["inner_needed"] / max(1, st["queries"]),
                "leaf_evals_per_step": (st["sphere_evals"] + st["box_evals"]) / max(1, st["steps"]),
                "leaf_needed_per_step": (st["sphere_needed"] + st["box_needed"]) / max(1, st["steps"]),
                "wave_inner_visits": st["wave_inner"], "wave_leaf_visits": st["wave_leaf"],
                "queries": st["queries"],
                "needed_over_issued_lane_slots": st["inner_needed"] / max(1, 64 * st["wave_inner"]),
                "list_steps_per_ray": st["list_steps"] / max(1, st["rays"]),
                "wave_list_passes": st["wave_lists"], "wave_list_evals": st["wave_list_evals"],
                "evals_per_list_pass": st["wave_list_evals"] / max(1, st["wave_lists"]),
                "lanes_advancing_per_iteration": st["steps"] / max(1, st["wave_steps"]),
            }
            if not a.no_variants:
                line["variants"] = variants(a, gpu, cam, mp, w, h, slots, depth, wl, dev)
            if not a.no_cpu:
                line["cpu_baseline"] = cpu_baseline(buf, cam, mp, w, h, a.cpu_seconds)
    knn = None
    if not a.no_knn:
        knn = knn_leg(a, rank, world, local_rank, dev, dist)
    if world > 1 and a.backend == "nccl" and not a.lib_comm and not a.no_lib_comm_leg:
        leg = lib_comm_leg(a, gpu, rank, world, dist, dev, cam, w, h, mp, slots, depth, line, knn)
        if rank == 0:
            line["lib_comm"] = leg
    if rank == 0:
        if knn is not None:
            line["knn"] = knn
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    gpu.close()


if __name__ == "__main__":
    main()
