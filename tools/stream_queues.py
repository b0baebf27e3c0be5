"""Which HIP streams of one process share a hardware queue?  A long spin kernel on stream i, a short kernel on stream j right
after it: j finishing first means the two ran at the same time (different queues).  Prints the sharing classes for the first K
streams created (normal priority, non-blocking), with the default GPU_MAX_HW_QUEUES and whatever the environment sets.
usage: python tools/stream_queues.py [K=10]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
x = torch.zeros(1, device=dev)
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(K)]
torch.cuda.synchronize()
spin = 20_000_000                     # ~10 ms
same = [[False] * len(streams) for _ in streams]
for i, si in enumerate(streams):
    for j, sj in enumerate(streams):
        if i == j:
            same[i][j] = True; continue
        ei, ej = torch.cuda.Event(), torch.cuda.Event()
        torch.cuda.synchronize()
        with torch.cuda.stream(si):
            torch.cuda._sleep(spin); ei.record(si)
        with torch.cuda.stream(sj):
            x.add_(1.0); ej.record(sj)
        ej.synchronize()
        same[i][j] = ei.query()        # the spin had finished by the time j's kernel did: j waited behind it
        torch.cuda.synchronize()
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
print("stream 0 = the current (null) stream, 1.. = created in order; X: j waits behind i")
for i in range(len(streams)):
    print(f"{i:2d} " + "".join("X" if same[i][j] else "." for j in range(len(streams))))
