"""Which torch streams run beside a given one?  (HIP deals streams onto a few hardware queues.)  Prints pair/single ratios of the
busy-wait probe for 16 pool streams, for event flavours and tick counts; diagnostic for the co-residency test's stream choice."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wavjepa_amd import ops

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
s_hold = torch.cuda.Stream()
ops.spin(100, stream=s_hold.cuda_stream)


def single(ticks):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s_hold); ops.spin(ticks, stream=s_hold.cuda_stream); e1.record(s_hold); e1.synchronize()
    return e0.elapsed_time(e1)


def pair(cand, ticks, timing, sync):
    best = 1e9
    for _ in range(3):
        go, done = torch.cuda.Event(enable_timing=timing), torch.cuda.Event(enable_timing=timing)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if sync:
            torch.cuda.synchronize()
        go.record(s_hold); cand.wait_event(go); e0.record(s_hold)
        ops.spin(ticks, stream=s_hold.cuda_stream); ops.spin(ticks, stream=cand.cuda_stream)
        done.record(cand); s_hold.wait_event(done); e1.record(s_hold); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


pool = [torch.cuda.Stream() for _ in range(16)]
for ticks in (30_000, 100_000):
    s = min(single(ticks) for _ in range(3))
    for timing in (True, False):
        for sync in (False, True):
            print(f"ticks {ticks} single {s:.3f} ms  timing_events={timing} sync={sync}:",
                  [round(pair(c, ticks, timing, sync) / s, 2) for c in pool], flush=True)
