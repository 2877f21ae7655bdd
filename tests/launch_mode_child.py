"""Child of tests/test_gpu_local.py::test_launch_mode_group: two ranks of one ncclCommInitAll communicator on cuda:0, each driven
by a thread of its own with NCCL_LAUNCH_MODE=GROUP (src/group.cc:128-190): three all-reduces per thread, the second thread
starting late; then one thread driving both ranks inside ncclGroupStart/End with two operations per communicator."""
import os
import sys
import threading
import time

os.environ["NCCL_LAUNCH_MODE"] = "GROUP"
os.environ["RCCL_AMD_MULTI_RANK_GPU"] = "1"
os.environ.setdefault("RCCL_AMD_SPIN_TIMEOUT_MS", "20000")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import rccl_amd as ra  # noqa: E402


def main():
    torch.cuda.set_device(0)
    n, count = 2, 100003
    comms = ra.Communicator.init_all(n, [0] * n)
    streams = [torch.cuda.Stream() for _ in range(n)]
    ins = [[torch.full((count,), float(10 * r + k + 1), device="cuda") for k in range(3)] for r in range(n)]
    outs = [[torch.zeros(count, device="cuda") for _ in range(3)] for _ in range(n)]
    torch.cuda.synchronize()
    errors = []

    def drive(r):
        try:
            torch.cuda.set_device(0)
            if r == 1:
                time.sleep(0.3)                      # the other thread has to wait at the launch barrier
            for k in range(3):
                comms[r].all_reduce(ins[r][k], outs[r][k], count, ra.ncclFloat32, ra.ncclSum, streams[r])
            streams[r].synchronize()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    th = [threading.Thread(target=drive, args=(r,)) for r in range(n)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert not errors and not any(t.is_alive() for t in th), errors
    assert time.time() - t0 >= 0.3
    for k in range(3):
        want = float(sum(10 * r + k + 1 for r in range(n)))
        for r in range(n):
            assert bool((outs[r][k] == want).all()), (r, k)
    # one thread, both ranks, two operations per communicator in one group: two rounds
    more = [[torch.zeros(count, device="cuda") for _ in range(2)] for _ in range(n)]
    ra.group_start()
    for k in range(2):
        for r in range(n):
            comms[r].all_reduce(ins[r][k], more[r][k], count, ra.ncclFloat32, ra.ncclSum, streams[r])
    ra.group_end()
    for s in streams:
        s.synchronize()
    for k in range(2):
        want = float(sum(10 * r + k + 1 for r in range(n)))
        for r in range(n):
            assert bool((more[r][k] == want).all()), (r, k)
    for c in comms:
        assert c.async_error() == 0
        c.destroy()
    print("launch mode group ok")


if __name__ == "__main__":
    main()
