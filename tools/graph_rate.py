#!/usr/bin/env python3
"""Runs ON THE GPU BOX: small batches (the launch-bound regime) as ordinary launches, as a recorded hipGraph of the same
launches, and as multi-cycle launches (filters resident in LDS between cycles).  filter-cycles/s and microseconds per cycle.

usage: python3 tools/graph_rate.py [precision=f64] [cycles=2000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import slam_pose_estimation_amd as spe  # noqa: E402


def main():
    prec = 1 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else 0
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    tdt = torch.float64 if prec == 0 else torch.float32
    s = spe.synth
    per_graph = 100
    print(f"# Pose, {'fp32' if prec else 'fp64'}, {cycles} fused cycles per figure; graph = {per_graph} launches recorded once and replayed; "
          f"multi8 = ukfb_cycle_multi_dev with 8 cycles per launch")
    print(f"{'filters':>8} {'launches us/cycle':>18} {'graph us/cycle':>15} {'multi8 us/cycle':>16} {'launches M/s':>13} {'graph M/s':>10} {'multi8 M/s':>11}")
    for n in (64, 256, 1024, 4096, 16384, 65536):
        mu, cov = s.pose_initial(n)
        acc, z, Q = s.pose_cycle_inputs(n, 0, mu[:, :3])
        a_t = torch.from_numpy(acc).to("cuda", tdt)
        z_t = torch.from_numpy(z).to("cuda", tdt)
        Q_t = torch.from_numpy(Q.reshape(n, 9)).to("cuda", tdt)
        z_r = torch.stack([z_t] * 2).contiguous()
        Q_r = torch.stack([Q_t] * 2).contiguous()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        e = spe.BatchPoseUKF(n, precision=prec, stream=int(side.cuda_stream))
        e.set_acceleration(None, 0.01 * np.eye(3))
        e.bind_acceleration_dev(a_t)

        def reset():
            e.initialize(mu, cov)
            e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
            e.sync()

        def timed(fn, reps):
            reset()
            fn()                       # warm
            e.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            e.sync()
            return (time.perf_counter() - t0)

        t_launch = timed(lambda: e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t), cycles) / cycles
        reset()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(per_graph):
                e.cycle_dev(0.01, spe.MEAS_POS3, z_t, Q_t)
        torch.cuda.synchronize()

        def replay():
            with torch.cuda.stream(side):
                graph.replay()
        t_graph = timed(replay, max(1, cycles // per_graph)) / (max(1, cycles // per_graph) * per_graph)
        t_multi = timed(lambda: e.cycle_multi_dev(8, 0.01, spe.MEAS_POS3, z_r, Q_r, 2, 0), max(1, cycles // 8)) / (max(1, cycles // 8) * 8)
        assert e.status_summary() == 0
        print(f"{n:>8} {t_launch * 1e6:>18.2f} {t_graph * 1e6:>15.2f} {t_multi * 1e6:>16.2f} {n / t_launch / 1e6:>13.1f} {n / t_graph / 1e6:>10.1f} {n / t_multi / 1e6:>11.1f}",
              flush=True)
        del graph
        e.close()


if __name__ == "__main__":
    main()
