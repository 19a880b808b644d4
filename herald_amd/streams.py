"""Choice of a side stream that runs BESIDE a given main stream.

HIP multiplexes streams onto a few hardware queues, and which queue a stream gets depends on how many streams the process has
used before.  Two findings of round 6 (docs/EXPERIMENTS.md, profiles/r06/cache_tier_third_instance.txt, stream_pairs.txt):
two streams of one priority may share a queue and then run in submission order; and for about one in four orders of stream
creation a pair of streams -- whatever their priorities -- is an UNLUCKY pair: whenever one of them has an event wait parked at
the head of its queue, the launches of the other take three times as long (7.5 -> 25 us for a 32 MB fill; the HET cache's row
launches 19 -> 56-96 us per pair).  Nothing in the HIP API tells the pairs apart, a measurement does: `pick_side_stream` creates a
few candidates and keeps the one beside which the main stream runs fastest under the pattern of a block-pipelined engine (the
side stream waits for the main stream's work of two blocks ago, the main stream for the side stream's of this block)."""
import os

import torch

_PICKED = {}


def _probe(main, side, buf, small, blocks=6, per=12):
    import ctypes
    from . import _lib
    L = _lib.load()
    booked = [torch.cuda.Event() for _ in range(blocks)]
    done = [torch.cuda.Event() for _ in range(blocks)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for b in range(blocks):
        with torch.cuda.stream(side):
            if b >= 2:
                side.wait_event(done[b - 2])
            # busy beside the main stream's block as a bookkeeping chain is: workgroups that hold LDS, then a few that stay
            # resident for most of the block (ha_debug_occupy: an occupant kernel of the library, no memory traffic)
            L.ha_debug_occupy(16, 1024, 96 * 1024, 2500, ctypes.c_void_p(side.cuda_stream))
            L.ha_debug_occupy(32, 256, 64, 5000, ctypes.c_void_p(side.cuda_stream))
            small.add_(1)
            booked[b].record(side)
        with torch.cuda.stream(main):
            if b == 2:
                e0.record(main)
            main.wait_event(booked[b])
            for i in range(per):
                buf.fill_(float(i))
            done[b].record(main)
    with torch.cuda.stream(main):
        e1.record(main)
    e1.synchronize()
    side.synchronize()
    return e0.elapsed_time(e1) / ((blocks - 2) * per)


def pick_side_stream(main, priority=0, candidates=4):
    """A stream of `priority` on main's device that is not an unlucky partner of `main` (torch.cuda.Stream or None = the
    current stream).  The answer is cached per (device, main stream, priority); HA_STREAM_CALIBRATE=0: no measurement."""
    if main is None:
        main = torch.cuda.current_stream()
    dev = main.device
    key = (dev.index, main.cuda_stream, priority)
    if key in _PICKED:
        return _PICKED[key]
    first = torch.cuda.Stream(device=dev, priority=priority)
    if os.environ.get("HA_STREAM_CALIBRATE", "1") == "0" or candidates <= 1:
        _PICKED[key] = first
        return first
    buf = torch.empty(8 << 20, dtype=torch.float32, device=dev)         # 32 MB: a fill of ~7 us
    small = torch.zeros(1 << 16, device=dev)
    torch.cuda.synchronize(dev)
    cands, times = [first], []
    for i in range(candidates):
        if i:
            cands.append(torch.cuda.Stream(device=dev, priority=priority))
        times.append(_probe(main, cands[i], buf, small))
        if i and times[i] < 1.5 * min(times) and times[0] < 1.5 * min(times):
            break           # the first candidate is fine and so is another one: nothing to tell apart
    best = min(range(len(times)), key=lambda i: times[i])
    pick = cands[0] if times[0] < 1.5 * times[best] else cands[best]
    torch.cuda.synchronize(dev)
    del buf, small
    _PICKED[key] = pick
    return pick
