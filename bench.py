#!/usr/bin/env python3
"""Headline benchmark: ncclAllReduce, sum, fp32, 1 GiB device-resident bucket.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; a step is one ncclAllReduce (through librccl_amd.so's C ABI) over one
synthetic random bucket already resident in HBM.  Rank 0 prints one JSON line:
  value      bus bandwidth in GB/s, (S/t) * 2(n-1)/n  (n = 1: the factor is 0, so the copy's
             algorithm bandwidth S/t is reported instead — src/device/onerank.cu is a copy)
  roofline   the dominant kernel's algorithmic bytes per launch / its average duration
             (HIP events on the launch stream) against the bound that applies:
             n = 1: HBM, 2*S bytes per launch, peak 8000 GB/s;
             n > 1, one rank per GPU: xGMI, 2(n-1)/n*S bytes leave each GPU per launch, peak (n-1) links x 153 GB/s;
             n > 1, several ranks on one GPU (loop-back, no link is used): HBM, the design's
             2S + 4(n-1)/n*S bytes per rank (2(n-1)/n*S less with a registered receive buffer) x ranks on the GPU
  cpu_baseline  rank 0, every n: the oracle's OpenMP element-wise sum of the n host buckets (regenerated
             from the same seeds) on the box's host cores, as S/t and with the same bus factor as `value`
             (a reported baseline, not a target)

torch is used for device memory, streams, events and the gloo rendezvous only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec peak
XGMI_LINK_GBPS = 153.0        # per xGMI link


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bytes", type=int, default=1 << 30, help="bucket size in bytes (default 1 GiB)")
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16", "fp16"])
    ap.add_argument("--coll", default="allreduce", choices=["allreduce", "reducescatter", "allgather", "broadcast", "reduce"])
    ap.add_argument("--in-place", action="store_true")
    ap.add_argument("--register", action="store_true",
                    help="ncclCommRegister the receive buffer first (zero-copy gather phase); off in the default run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    import rccl_amd as ra

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    n = world
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py needs a GPU")
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    per_gpu = (n + ndev - 1) // ndev
    if per_gpu > 1:
        # Loop-back mode (several ranks on one GPU, for kernel work on a one-GPU box): the ranks share
        # 256 CUs, so cap the channels to keep every workgroup of every rank resident.
        # Workgroups per channel: the ring has one, the chain two (up and down), the all-pairs schedule three for an
        # all-reduce with three ranks and more and two otherwise (two ranks; reduce-scatter and all-gather).
        # (two ranks without NCCL_ALGO: the library takes the ring from 24 MiB, reduce-scatter / all-gather from 16 MiB per rank)
        algo = os.environ.get("NCCL_ALGO", "").lower()
        two_rank_ring = (n == 2 and not algo and os.environ.get("RCCL_AMD_TWO_RANK_RING", "1") != "0"
                         and not os.environ.get("NCCL_P2P_DISABLE")
                         and ((args.coll == "allreduce" and args.bytes >= 24 << 20)
                              or (args.coll in ("reducescatter", "allgather") and args.bytes >= 16 << 20)))
        fused = os.environ.get("RCCL_AMD_FUSED_ROLES", "0") != "0" and n > 2 and args.coll == "allreduce" and algo in ("", "direct")
        wgs = 1 if (algo == "ring" or two_rank_ring or fused) else 2 if (algo == "tree" or n == 2 or args.coll != "allreduce") else 3
        # A launch's workgroups are dealt round-robin to the 8 XCDs (32 compute units each, one 1024-thread workgroup per
        # compute unit), so what has to fit is ranks x ceil(workgroups of one rank / 8) <= 32 per XCD: 3 ranks x 84
        # workgroups is 252 in total but 33 on four of the XCDs; the 33rd waits for a compute unit that only frees up when
        # a channel has finished (1 GiB at 3 ranks: 3.32 ms with 26 channels, 4.12 ms with 28; 7 ranks x 12 channels hung).
        os.environ.setdefault("RCCL_AMD_NCHANNELS", str(max(2, 8 * (32 // per_gpu) // wgs)))

    out = sys.stdout
    if n > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")      # a failing rank says why on stderr
        # gloo reports its connections on file descriptor 1: keep the real stdout for the one JSON line and send
        # everything else to stderr
        out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=n)

    def barrier():
        if n > 1:
            dist.barrier()

    dt_code = {"fp32": ra.ncclFloat32, "bf16": ra.ncclBfloat16, "fp16": ra.ncclFloat16}[args.dtype]
    tdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    es = ra.TYPE_SIZE[dt_code]
    S = args.bytes                                   # bytes per rank (all-reduce) / per rank block
    count = S // es

    def agree(err):
        """Every rank learns whether any rank failed; rank 0 then prints a line that says why and all leave."""
        if n == 1:
            if err:
                sys.exit(err)
            return
        errs = [None] * n
        dist.all_gather_object(errs, err)
        if any(errs):
            if rank == 0:
                print(json.dumps({"metric": "AllReduce bus-bandwidth GB/s, 1 GiB fp32 sum", "value": None, "unit": "GB/s",
                                  "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
                                  "error": {str(r): e for r, e in enumerate(errs) if e}}), file=out, flush=True)
            sys.exit(1)

    # communicator: unique id from rank 0 over gloo (what ncclGetUniqueId is for)
    err, comm = None, None
    try:
        if n > 1:
            uid = [ra.get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = ra.Communicator.init_rank(n, uid[0], rank)
        else:
            comm = ra.Communicator.init_all(1, [dev])[0]
    except ra.NcclError as e:
        err = f"communicator set-up on device {dev}: {e}"
    agree(err)

    # synthetic random bucket, seed 0x5EED0000 + rank, uniform [-1, 1)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5EED0000 + rank)
    if args.coll in ("allreduce", "broadcast", "reduce"):
        in_elems, out_elems = count, count
    elif args.coll == "reducescatter":
        in_elems, out_elems = count * n, count
    else:
        in_elems, out_elems = count, count * n
    sendbuf = (torch.rand(in_elems, device="cuda", generator=gen, dtype=torch.float32) * 2 - 1).to(tdt)
    recvbuf = sendbuf if (args.in_place and args.coll == "allreduce") else torch.zeros(out_elems, device="cuda", dtype=tdt)
    stream = torch.cuda.current_stream()
    reg_handle = None
    recv_arg, recv_raw = recvbuf, None
    send_arg, send_raw = sendbuf, None
    if args.register and n > 1 and args.coll == "reducescatter":
        # the peers' reducers read a registered SEND buffer directly: it has to be fine-grained memory too
        import ctypes
        send_raw = ra.mem_alloc(in_elems * es)
        rc = ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(send_raw), ctypes.c_void_p(sendbuf.data_ptr()),
                                                     ctypes.c_size_t(in_elems * es), 3)
        if rc != 0:
            sys.exit(f"rank {rank}: hipMemcpy into the registered send buffer failed ({rc})")
        send_arg = send_raw
        reg_handle = comm.register(send_raw, in_elems * es)
        barrier()
    elif args.register and n > 1:
        # peers write only into fine-grained memory: the registered receive buffer comes from ncclMemAlloc (the launches
        # get its address; it is copied into `recvbuf` for the checks afterwards)
        if not (args.in_place and args.coll == "allreduce"):
            recv_raw = ra.mem_alloc(out_elems * es)
            recv_arg = recv_raw
        reg_handle = comm.register(recv_raw if recv_raw else recvbuf.data_ptr(), out_elems * es)
        barrier()                                       # every rank's registration is published before the first launch

    def step():
        if args.coll == "allreduce":
            comm.all_reduce(sendbuf, recv_arg, count, dt_code, ra.ncclSum, stream)
        elif args.coll == "reducescatter":
            comm.reduce_scatter(send_arg, recv_arg, count, dt_code, ra.ncclSum, stream)
        elif args.coll == "broadcast":
            comm.broadcast(sendbuf, recv_arg, count, dt_code, 0, stream)
        elif args.coll == "reduce":
            comm.reduce(sendbuf, recv_arg, count, dt_code, ra.ncclSum, 0, stream)
        else:
            comm.all_gather(sendbuf, recv_arg, count, dt_code, stream)

    try:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if comm.async_error() != 0:
            err = f"asynchronous error {comm.async_error()} during warm-up: {ra.lib().ncclGetLastError(None).decode()}"
    except (ra.NcclError, RuntimeError) as e:
        err = f"warm-up: {e}"
    agree(err)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    wall = t1 - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps      # average launch duration on the launch stream
    if comm.async_error() != 0:
        sys.exit(f"rank {rank}: the communicator reported an asynchronous error")
    if recv_raw:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(ctypes.c_void_p(recvbuf.data_ptr()), ctypes.c_void_p(recv_raw), ctypes.c_size_t(out_elems * es), 3)
        if rc != 0:
            sys.exit(f"rank {rank}: hipMemcpy of the registered receive buffer failed ({rc})")

    if n > 1:
        t = torch.tensor([wall, kern_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(t[0]), float(t[1])

    # Correctness of the timed path's last output (all-reduce), two ways:
    #  - 64 windows of 4096 elements spread evenly over the whole bucket, the last one ending at the last element (every
    #    channel and round of the schedule is met): every rank's windows are gathered over gloo and folded by the oracle;
    #  - the WHOLE receive buffer on every rank against a fold computed on the device in rank order from the ranks'
    #    buckets, which each rank regenerates from their seeds (a property check next to the oracle's: the number of
    #    elements that differ must be zero on every rank).
    verified = None
    heads = None
    whole_buffer_mismatches = None
    sample_idx = None
    if args.coll == "allreduce" and not args.in_place and args.dtype == "fp32":
        import oracle as orc
        nwin, wlen = 64, min(4096, count)
        starts = [(count - wlen) * k // (nwin - 1) for k in range(nwin)]
        sample_idx = torch.cat([torch.arange(s, s + wlen, device="cuda") for s in starts])
        head = sendbuf[sample_idx].float().cpu().numpy()
        if n > 1:
            heads = [None] * n
            dist.all_gather_object(heads, head)
        else:
            heads = [head]
        want = orc.host_fold(orc.F32, orc.SUM, heads)
        got = recvbuf[sample_idx].float().cpu().numpy()
        # the all-pairs schedule folds in rank order like the host fold: same bits; ring and chain orders differ
        # in the last place, so they are held to the reference's fp32 tolerance instead
        direct = n == 1 or comm.plan_info(0, count, dt_code)["algo"] == 0
        verified = bool(np.array_equal(got, want)) if direct else bool(orc.first_mismatch(orc.F32, got, want) < 0)
        acc = None
        for r in range(n):
            if r == rank:
                b = sendbuf
            else:
                g2 = torch.Generator(device="cuda")
                g2.manual_seed(0x5EED0000 + r)
                b = torch.rand(count, device="cuda", generator=g2, dtype=torch.float32) * 2 - 1
            acc = b.clone() if acc is None else acc.add_(b)
            del b
        if direct:
            bad = int((acc != recvbuf).sum().item())
        else:
            bad = int(((acc - recvbuf).abs() > n * n * 2.0 ** -24).sum().item())    # n - 1 roundings of sums below n
        del acc
        if n > 1:
            tb = torch.tensor([bad], dtype=torch.int64)
            dist.all_reduce(tb, op=dist.ReduceOp.SUM)
            bad = int(tb[0])
        whole_buffer_mismatches = bad
        verified = verified and bad == 0

    per_step = wall / args.steps
    if args.coll == "allreduce":
        algbw = S / per_step / 1e9
        factor = 2.0 * (n - 1) / n
        link_bytes = factor * S
    elif args.coll in ("broadcast", "reduce"):
        algbw = S / per_step / 1e9
        factor = 1.0
        link_bytes = (n - 1) * S                       # what leaves (or reaches) the root
    else:
        total = S * n
        algbw = total / per_step / 1e9
        factor = (n - 1) / n
        link_bytes = (n - 1) * S
    busbw = algbw * factor if n > 1 else algbw

    per_gpu = (n + ndev - 1) // ndev
    traffic_source = None
    if n == 1:
        bound, peak = "hbm", HBM_PEAK_GBPS
        alg_bytes = 2.0 * S                            # read the bucket once, write it once
        bound_note = "one rank: device copy, 2*S bytes per launch"
    elif per_gpu > 1:
        # Loop-back: the ranks of this run share GPUs, no byte crosses a link, HBM is the bound.  Bytes the design moves
        # through HBM per rank and launch: input read and output written once, every link byte written into a window
        # and read back once per phase (all-reduce: two phases; one when the gather phase is written directly).
        bound, peak = "hbm", HBM_PEAK_GBPS
        if args.coll == "allreduce":
            per_rank = 2.0 * S + (2.0 if reg_handle else 4.0) * (n - 1) / n * S
        elif args.coll in ("broadcast", "reduce"):
            per_rank = 2.0 * S + 2.0 * (n - 1) / n * S      # averaged over ranks: the root's n-1 copies, each written and read
        else:
            per_rank = (n + 1.0) * S + 2.0 * (n - 1) * S     # (n+1)*S floor plus window write + read of n-1 blocks
        alg_bytes = per_rank * per_gpu
        bound_note = f"{per_gpu} ranks share each GPU: HBM bytes of the design per GPU and launch, no xGMI traffic"
    else:
        bound, peak = "xgmi", (n - 1) * XGMI_LINK_GBPS
        alg_bytes = link_bytes                         # bytes that leave each GPU per launch
        bound_note = f"one rank per GPU: bytes leaving each GPU per launch over {n - 1} links"
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            with open(tfile) as f:
                traffic = json.load(f).get(f"{args.coll}_{args.dtype}_{S}_n{n}")
            if traffic is not None:
                traffic_source = ("profiles/traffic.json: rocprofv3 --pmc passes of this command, committed with the "
                                  "profile they came from; not re-measured in this run")
        except (OSError, ValueError):
            traffic = None

    line = {
        "metric": ("AllReduce bus-bandwidth GB/s, 1 GiB fp32 sum" if (args.coll == "allreduce" and args.dtype == "fp32" and S == 1 << 30)
                   else f"{args.coll} bus-bandwidth GB/s, {S} B per rank, {args.dtype}"),
        "value": round(busbw, 3), "unit": "GB/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(per_step * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": "bf16", "fp16": "f16"}[args.dtype], "data": "synthetic",
        "config": {"workload": f"nccl{'AllReduce' if args.coll == 'allreduce' else args.coll} sum {args.dtype} "
                               f"{S} B per rank, out-of-place, device-resident random bucket"
                               + ("" if n > 1 else " (1 rank: the reference path is a device copy; value is its S/t)"),
                   "ranks": n, "ranks_per_gpu": (n + ndev - 1) // ndev, "algbw_GBps": round(algbw, 3),
                   "registered_recvbuff": bool(reg_handle), "gather_phase_pieces": comm.stats() if n > 1 else None,
                   # RCCL_AMD_TUNE_FLAGS=32: thread-0 stamps around the phases of every protocol step, averaged per role
                   "step_timers_rank0": (comm.step_timers() or None) if n > 1 else None,
                   # NCCL_P2P_DISABLE=1: rank 0's proxy threads (tx / rx) since the communicator was made: ms inside send() and
                   # recv() for slot data, ms idle / in poll(), messages, payload bytes
                   "host_staged_proxy_rank0": (comm.net_stats() or None) if n > 1 else None,
                   "plan": comm.plan_info({"allreduce": 0, "reducescatter": 1, "allgather": 2, "broadcast": 3, "reduce": 4}[args.coll],
                                          count, dt_code) if n > 1 else {"algo": "local copy"}},
        "verified_against_oracle": verified,
        "verification": {"oracle_windows": "64 x 4096 elements spread over the bucket, tail included",
                         "whole_buffer_mismatches_all_ranks": whole_buffer_mismatches} if verified is not None else None,
        "roofline": {"bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": "GB/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms": round(kern_ms, 4), "algorithmic_bytes_per_launch": alg_bytes, "bound_note": bound_note},
    }

    if rank == 0 and not args.no_cpu_baseline and args.coll == "allreduce" and args.dtype == "fp32":
        import oracle as orc
        # the n host buckets: rank 0 regenerates every rank's bucket from its seed on its own GPU (same generator,
        # same device model, same values) and keeps host copies; the sampled windows are compared with what the
        # ranks themselves hold
        hosts = []
        same = True
        for r in range(n):
            if r == 0:
                hosts.append(sendbuf.cpu().numpy())
            else:
                g2 = torch.Generator(device="cuda")
                g2.manual_seed(0x5EED0000 + r)
                t = torch.rand(count, device="cuda", generator=g2, dtype=torch.float32) * 2 - 1
                hosts.append(t.cpu().numpy())
                del t
            if heads is not None:
                same = same and bool(np.array_equal(hosts[r][sample_idx.cpu().numpy()], heads[r]))
        host_out = np.empty_like(hosts[0])
        orc.omp_sum_f32(hosts, host_out)                   # page in
        reps, t_cpu = 0, 0.0
        while t_cpu < args.cpu_seconds and reps < 50:
            c0 = time.perf_counter()
            orc.omp_sum_f32(hosts, host_out)
            t_cpu += time.perf_counter() - c0
            reps += 1
        cpu_algbw = S * reps / t_cpu / 1e9
        cpu_value = cpu_algbw * factor if n > 1 else cpu_algbw
        line["cpu_baseline"] = {"value": round(cpu_value, 3), "unit": "GB/s", "cores": orc.omp_threads(), "kind": "port",
                                "algbw_GBps": round(cpu_algbw, 3), "bus_factor": factor if n > 1 else None,
                                "sample": f"{reps} passes of the OpenMP element-wise sum (oracle/oracle.c, "
                                          f"'#pragma omp parallel for simd') of {n} host bucket(s) of {S} B fp32 into one "
                                          f"output: reads {n}*S, writes S per pass; value = S/t"
                                          + (f" x 2(n-1)/n = {factor:.3f} (the bus factor of `value`)" if n > 1
                                             else " (one rank: a host copy, as the device path is a device copy)"),
                                "buckets_match_ranks": same if heads is not None else None,
                                "host_cpus": os.cpu_count()}

    if reg_handle:
        comm.deregister(reg_handle)
    if recv_raw:
        ra.mem_free(recv_raw)
    if send_raw:
        ra.mem_free(send_raw)
    comm.destroy()
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), file=out, flush=True)


if __name__ == "__main__":
    main()
