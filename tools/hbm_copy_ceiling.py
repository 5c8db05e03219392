"""GPU box: what a plain device copy reaches on this MI355X at the GAE scan's streaming footprint (1.1 GB moved per launch, about
half read / half written) -- the practical ceiling to read the scan's HBM fraction against.  torch.Tensor.copy_ (vectorised
elementwise kernel) and hipMemcpyDtoD through torch; bytes moved = read + written."""
import json, torch
dev = torch.device("cuda:0")
out = {}
for mb in (8.65, 69.2, 553.6):       # read size in MB = half of the scan's 33 B/element: headline, 32 768 envs, 262 144 envs
    n = int(mb * 1000 * 1000) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    out[f"copy_{mb}MB_read_plus_{mb}MB_written"] = {"us": round(us, 2), "GB_per_s": round(2 * n * 4 / us / 1e3, 1),
                                                     "frac_of_8TBps": round(2 * n * 4 / us / 1e3 / 8000, 4)}
print(json.dumps(out))
