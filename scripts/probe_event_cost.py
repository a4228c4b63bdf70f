"""what an event record / cross-stream wait costs on the stream that is recorded (the input-gradient chain of the reverse pass records one
per layer): N short kernels back to back on stream A, (a) alone, (b) with hipEventRecord between them and stream B waiting on each,
(c) the same with hipEventDisableTiming | hipEventReleaseToDevice events, (d) hipStreamWriteValue32 on A / hipStreamWaitValue32 on B
instead of events.  usage: python scripts/probe_event_cost.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
N = 200
a = torch.zeros(1 << 16, device=dev)
b = torch.zeros(1 << 16, device=dev)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def run(kind):
    evs = []
    if kind in ("event", "event_dev"):
        for _ in range(N):
            e = ctypes.c_void_p()
            flags = 0x2 if kind == "event" else (0x2 | 0x40000000)
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
            evs.append(e)
    sig = None
    if kind == "value":
        sig = ctypes.c_void_p()
        assert hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2) == 0      # hipMallocSignalMemory
        assert hip.hipMemset(sig, 0, 8) == 0
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sA):
        t0.record(sA)
        for k in range(N):
            a.add_(1.0)
            if kind in ("event", "event_dev"):
                assert hip.hipEventRecord(evs[k], ctypes.c_void_p(sA.cuda_stream)) == 0
                assert hip.hipStreamWaitEvent(ctypes.c_void_p(sB.cuda_stream), evs[k], 0) == 0
            elif kind == "value":
                assert hip.hipStreamWriteValue32(ctypes.c_void_p(sA.cuda_stream), sig, k + 1, 0) == 0
                assert hip.hipStreamWaitValue32(ctypes.c_void_p(sB.cuda_stream), sig, k + 1, 0, 0xFFFFFFFF) == 0   # hipStreamWaitValueGte
            if kind != "alone":
                with torch.cuda.stream(sB):
                    b.add_(1.0)
        t1.record(sA)
    torch.cuda.synchronize()
    for e in evs:
        hip.hipEventDestroy(e)
    return t0.elapsed_time(t1) * 1e3 / N


for _ in range(2):
    for kind in ("alone", "event", "event_dev", "value"):
        try:
            print(f"{kind:10s} {run(kind):6.2f} us per kernel on the recorded stream")
        except Exception as ex:
            print(kind, "failed:", ex)
