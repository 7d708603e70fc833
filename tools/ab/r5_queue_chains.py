"""Experiment (round 5): what does a chain of tiny dependent kernels cost when S streams run such chains at once?  (The 5-us kernels of a small proof's MSM
tails and witness map show as 50-160 us in rocprofv3 timelines while three queues are active.)  torch elementwise kernels on 256-element tensors."""
import os, sys, time
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
for S in (1, 2, 3, 4, 6, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    xs = [torch.zeros(256, device=dev) for _ in range(S)]
    def run():
        for i in range(N):
            for s, x in zip(streams, xs):
                with torch.cuda.stream(s):
                    x.add_(1.0)
    run(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); t_issue = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = min(best, dt)
    print(f"S={S} streams x {N} dependent tiny kernels: {best * 1e6 / N:.2f} us per chain step (issue {t_issue * 1e6 / N:.2f} us), GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}", flush=True)
