import sys, torch
sys.path.insert(0, ".")
from clsr_amd import ops
def t(nbytes):
    a, b = torch.empty(nbytes // 4, device="cuda"), torch.empty(nbytes // 4, device="cuda")
    f = lambda: ops.call("clsr_copy_words", b, a.data_ptr(), nbytes)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); e1.synchronize()
    return 2 * nbytes * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
print(" ".join("%dMB:%.0f" % (n >> 20, t(n)) for n in (64 << 20, 256 << 20, 1 << 30, 4 << 30)))
