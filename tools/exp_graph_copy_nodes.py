"""After the captured-memset finding: are the OTHER non-kernel nodes a capture of this code base can contain replayed correctly?
Memcpy nodes -- device to device (torch's tensor.copy_ of contiguous tensors), device to pinned host and pinned host to device (non_blocking) --
first in the graph and behind a kernel, small and large, four replays each with the source changed before every replay.
    python tools/exp_graph_copy_nodes.py"""
import torch

dev = torch.device("cuda:0")
x = torch.zeros(16, device=dev)
bad = 0
for n in (1, 4, 1024, 1 << 20, 17 << 20):
    for order in ("first", "mid"):
        src = torch.zeros(n, dtype=torch.float32, device=dev)
        dst = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
        host_out = torch.full((n,), -1.0, dtype=torch.float32).pin_memory()
        host_in = torch.zeros(n, dtype=torch.float32).pin_memory()
        dst2 = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            if order == "mid":
                x.add_(1.0)
            dst.copy_(src)
            host_out.copy_(src, non_blocking=True)
            dst2.copy_(host_in, non_blocking=True)
            x.add_(1.0)
        res = []
        for rep in range(4):
            v = float(rep + 1)
            src.fill_(v); host_in.fill_(v + 0.5)
            dst.fill_(-1.0); dst2.fill_(-1.0); host_out.fill_(-1.0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            ok = (bool((dst == v).all()), bool((host_out == v).all()), bool((dst2 == v + 0.5).all()))
            res.append(ok)
            bad += sum(not o for o in ok)
        print(f"n {n:9d} {order:5s} (d2d, d2h, h2d) ok per replay: {res}", flush=True)
print("all copies replayed correctly" if bad == 0 else f"{bad} WRONG COPIES")
