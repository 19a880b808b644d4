"""What a plain device copy of the step's byte count costs launch to launch (development aid): the floor a single ~42 MB
launch can reach on this part, beside the step's 12.3 us.  `resident`: the same buffers every launch (they stay in the
256 MB Infinity Cache); `from HBM`: every launch a fresh 1/192 of two 4 GiB buffers, as the step's rows, gradients and
outputs are."""
import torch
dev = torch.device("cuda:0")
big_a = torch.empty(1 << 30, dtype=torch.float32, device=dev)       # 4 GiB
big_b = torch.empty(1 << 30, dtype=torch.float32, device=dev)
big_a.normal_()
for mb in (10.7, 21.4, 42.8, 85.6, 342.4):
    n = int(mb * 1e6 / 2 / 4)            # read n floats + write n floats = mb MB of traffic
    for mode in ("resident", "from HBM"):
        nchunk = 1 if mode == "resident" else (1 << 30) // n
        srcs = [big_a[i * n:(i + 1) * n] for i in range(nchunk)]
        dsts = [big_b[i * n:(i + 1) * n] for i in range(nchunk)]
        reps = 384
        for i in range(32):
            dsts[i % nchunk].copy_(srcs[i % nchunk])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            dsts[i % nchunk].copy_(srcs[i % nchunk])
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("copy moving %6.1f MB (read + write), %-9s: %6.2f us per launch = %.2f TB/s" % (mb, mode, us, mb / us))
